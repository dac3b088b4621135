#!/bin/bash
# GPU box, round 3: soak of the round's inflate (literal pairs; tokens on and off) and of the device / host front ends end to end.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03m
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
INQ_SOAK_SEED=31000 timeout -k 10 700 python3 tools/soak_inflate.py 60 4 > $OUT/soak_inflate.txt 2>&1; echo "soak_inflate rc $?" | tee -a $OUT/soak_inflate.txt; tail -6 $OUT/soak_inflate.txt
timeout -k 10 400 python3 tools/soak_e2e.py --cases 90 --frontend device --seed0 33000 > $OUT/soak_e2e_device.txt 2>&1; echo "soak_e2e device rc $?" | tee -a $OUT/soak_e2e_device.txt; tail -2 $OUT/soak_e2e_device.txt
timeout -k 10 300 python3 tools/soak_e2e.py --cases 40 --frontend host --seed0 34000 > $OUT/soak_e2e_host.txt 2>&1; echo "soak_e2e host rc $?" | tee -a $OUT/soak_e2e_host.txt; tail -2 $OUT/soak_e2e_host.txt
