#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04_state2
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests.txt 2>&1; echo "gpu tests rc $?"; tail -3 $OUT/gpu_tests.txt
bash tools/runs/r04_bench.sh 1100
cp $ROOT/gpurun_out/r04_bench/bench_default.json $OUT/bench_default.json
