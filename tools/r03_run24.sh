#!/bin/bash
# GPU box, round 3: shapes of the workgroup inflate that put more waves on a CU (192 lanes x 5 waves per SIMD; 2 KB of roots, 10 workgroups per CU).
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03s2
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
for lv in 1 6; do for k in cigar ont qual; do
  for lib in base t192 cap2k; do
    echo -n "level $lv $k $lib: " | tee -a $OUT/inflate_more_waves_per_cu.txt
    INQ_LIB=$ROOT/inquistr_amd/lib/libinq_$lib.so ALGO=0 timeout -k 10 200 python3 tools/inflate_bench.py 20000 $lv $k 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/comp .* kernel/kernel/' | tee -a $OUT/inflate_more_waves_per_cu.txt
  done
done; done
