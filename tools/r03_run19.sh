#!/bin/bash
# GPU box, round 3: lanes with nothing to repeat decode the next round ahead - tests and A/B.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03a3
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_front.py -m gpu -x -q > $OUT/gputest.log 2>&1; echo "pytest rc $?" | tee -a $OUT/gputest.log; tail -4 $OUT/gputest.log
for lv in 1 6; do for k in cigar seq ont qual; do
  for lib in base new; do
    echo -n "level $lv $k $lib: " | tee -a $OUT/inflate_next_round_decoded_ahead.txt
    L=""; [ $lib = base ] && L=$ROOT/inquistr_amd/lib/libinq_base.so
    INQ_LIB=$L ALGO=0 timeout -k 10 200 python3 tools/inflate_bench.py 20000 $lv $k 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/comp .* kernel/kernel/' | tee -a $OUT/inflate_next_round_decoded_ahead.txt
  done
done; done
