#!/bin/bash
# GPU box, round 3: CIGAR gather with 16 bytes per lane - front-end and end-to-end tests, kernel time in the CLI profile.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03cg
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_front.py tests/test_gpu_end_to_end.py -m gpu -x -q > $OUT/gputest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 $OUT/gputest.log
[ $rc = 0 ] || exit 1
timeout -k 10 400 bash tools/profile_front.sh r03_front_gather 50000 > $OUT/profile_front.log 2>&1; echo "profile_front rc $?"
head -6 $ROOT/gpurun_out/prof_r03_front_gather/kernel_stats.csv | cut -c1-150
