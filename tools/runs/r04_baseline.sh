#!/bin/bash
# GPU box, round 4 start: the state the round begins from (tests, allocation probe, default bench line).
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04_base
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout -k 10 60 inquistr_amd/lib/alloc_probe > $OUT/alloc_probe.txt 2>&1; echo "alloc_probe rc $?"
cat $OUT/alloc_probe.txt
timeout -k 10 60 inquistr_amd/lib/hip_startup_probe > $OUT/hip_startup.txt 2>&1; cat $OUT/hip_startup.txt
timeout -k 10 420 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $?"
tail -c 3000 $OUT/bench_default.json
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests.txt 2>&1; echo "gpu tests rc $?"; tail -3 $OUT/gpu_tests.txt
