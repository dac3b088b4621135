#!/bin/bash
# GPU box, round 3: long soak of the round's final kernels (new seeds): damaged deflate streams, random BAMs through both front ends.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03long
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
INQ_SOAK_SEED=424200 timeout -k 10 500 python3 tools/soak_inflate.py 400 3 > $OUT/soak_inflate.txt 2>&1; echo "soak_inflate rc $?" | tee -a $OUT/soak_inflate.txt; tail -2 $OUT/soak_inflate.txt
timeout -k 10 420 python3 tools/soak_e2e.py --cases 600 --frontend device --seed0 515000 > $OUT/soak_e2e_device.txt 2>&1; echo "soak_e2e device rc $?" | tee -a $OUT/soak_e2e_device.txt; tail -2 $OUT/soak_e2e_device.txt
timeout -k 10 120 python3 tools/soak_e2e.py --cases 100 --frontend host --seed0 616000 > $OUT/soak_e2e_host.txt 2>&1; echo "soak_e2e host rc $?" | tee -a $OUT/soak_e2e_host.txt; tail -2 $OUT/soak_e2e_host.txt
