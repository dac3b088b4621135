#!/bin/bash
# GPU box: the state of the round - inflate rates by kind of data (options at their defaults, and tokens forced on for the CIGAR-only
# kind), the full GPU suite, the default bench line.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04_state
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
for n in 20000 40000; do for lv in 1 6; do for k in cigar seq ont qual; do
  echo -n "$n blocks level $lv $k: " | tee -a $OUT/inflate_state.txt
  timeout -k 10 200 python3 tools/inflate_bench.py $n $lv $k 2>&1 | grep -v amdgpu.ids | grep kernel | tail -1 | tee -a $OUT/inflate_state.txt
done; done; done
for lv in 1 6; do
  echo -n "40000 blocks level $lv cigar, inflate_tokens = 1: " | tee -a $OUT/inflate_state.txt
  TOKENS=1 timeout -k 10 200 python3 tools/inflate_bench.py 40000 $lv cigar 2>&1 | grep kernel | tail -1 | tee -a $OUT/inflate_state.txt
done
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests.txt 2>&1; echo "gpu tests rc $?"; tail -3 $OUT/gpu_tests.txt
bash tools/runs/r04_bench.sh 1100
cp $ROOT/gpurun_out/r04_bench/bench_default.json $OUT/bench_default.json
