#!/bin/bash
# GPU box, round 3: around the 64-segment stretch: 2 KB of roots with it (ten workgroups per CU), 48 segments.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03k2
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
for lv in 1 6; do for k in cigar seq ont qual; do
  for lib in base cap2k s48; do
    echo -n "level $lv $k $lib: " | tee -a $OUT/inflate_around_the_64_segment_stretch.txt
    INQ_LIB=$ROOT/inquistr_amd/lib/libinq_$lib.so ALGO=0 timeout -k 10 200 python3 tools/inflate_bench.py 20000 $lv $k 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/comp .* kernel/kernel/' | tee -a $OUT/inflate_around_the_64_segment_stretch.txt
  done
done; done
