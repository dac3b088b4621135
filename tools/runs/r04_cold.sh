#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04_cold
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout -k 10 300 python3 tools/cold_vs_reread.py 100000 2>&1 | tee $OUT/cold_vs_reread_100k.txt
timeout -k 10 300 python3 tools/cold_vs_reread.py 50000 2>&1 | tee $OUT/cold_vs_reread_50k.txt
timeout -k 10 300 inquistr_amd/lib/hbm_read_peak 2>&1 | tee $OUT/hbm_read_peak.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "million" > $OUT/gpu_tests_million.txt 2>&1; echo "million tests rc $?"; tail -3 $OUT/gpu_tests_million.txt
