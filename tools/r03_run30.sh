#!/bin/bash
# GPU box, round 3: full GPU suite on the round's last state (host: resident server, kept target list, lazily loaded library).
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03final
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $OUT/gputest_full.log 2>&1; rc=$?; echo "pytest rc $rc" | tee -a $OUT/gputest_full.log; tail -3 $OUT/gputest_full.log
[ $rc = 0 ] || exit 1
timeout -k 10 120 python3 -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $OUT/smoke.log
timeout -k 10 300 python3 tools/soak_e2e.py --cases 120 --frontend device --seed0 99000 > $OUT/soak_e2e_device.txt 2>&1; echo "soak_e2e device rc $?"; tail -1 $OUT/soak_e2e_device.txt
