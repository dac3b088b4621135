#!/bin/bash
# GPU box, round 3: soak of the round's last kernels - random BAMs through both front ends, damaged deflate streams, new seeds.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03soak2
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
INQ_SOAK_SEED=131000 timeout -k 10 500 python3 tools/soak_inflate.py 120 3 > $OUT/soak_inflate.txt 2>&1; echo "soak_inflate rc $?" | tee -a $OUT/soak_inflate.txt; tail -3 $OUT/soak_inflate.txt
timeout -k 10 400 python3 tools/soak_e2e.py --cases 200 --frontend device --seed0 77000 > $OUT/soak_e2e_device.txt 2>&1; echo "soak_e2e device rc $?" | tee -a $OUT/soak_e2e_device.txt; tail -2 $OUT/soak_e2e_device.txt
timeout -k 10 200 python3 tools/soak_e2e.py --cases 60 --frontend host --seed0 78000 > $OUT/soak_e2e_host.txt 2>&1; echo "soak_e2e host rc $?" | tee -a $OUT/soak_e2e_host.txt; tail -2 $OUT/soak_e2e_host.txt
