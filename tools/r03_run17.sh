#!/bin/bash
# GPU box, round 3: long soak of the final inflate (forms chosen by the data) and of both front ends.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03r
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
INQ_SOAK_SEED=52000 timeout -k 10 800 python3 tools/soak_inflate.py 170 3 > $OUT/soak_inflate.txt 2>&1; echo "soak_inflate rc $?" | tee -a $OUT/soak_inflate.txt; tail -4 $OUT/soak_inflate.txt
timeout -k 10 300 python3 tools/soak_e2e.py --cases 160 --frontend device --seed0 56000 > $OUT/soak_e2e_device.txt 2>&1; echo "soak_e2e device rc $?" | tee -a $OUT/soak_e2e_device.txt; tail -2 $OUT/soak_e2e_device.txt
