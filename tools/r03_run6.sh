#!/bin/bash
# GPU box, round 3: literal runs in the inflate's symbol loop (pairs, triples, predictor) - correctness and A/B of five builds.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03g
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_front.py -m gpu -x -q > $OUT/gputest.log 2>&1; echo "pytest rc $?" | tee -a $OUT/gputest.log; tail -5 $OUT/gputest.log
for lv in 1 6; do
  for k in cigar ont qual seq; do
    for lib in libinq_nopair.so libinq_pair.so libinq_pairpred.so libinq_triple_nopred.so libinquistr_hip.so; do
      echo -n "level $lv $k $lib: " | tee -a $OUT/inflate_ab.txt
      INQ_LIB=$ROOT/inquistr_amd/lib/$lib ALGO=0 timeout -k 10 200 python3 tools/inflate_bench.py ${1:-20000} $lv $k 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/comp .* kernel/kernel/' | tee -a $OUT/inflate_ab.txt
    done
  done
done
