#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04_cigpause
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
D=/tmp/inq_r04; mkdir -p $D
( time timeout -k 10 300 python3 tools/make_synth_bam.py unphased100k 100000 $D/cig native 6 ) 2>&1 | grep -E "wrote|real"
echo "--- back to back" | tee $OUT/cig_runs.txt
timeout -k 10 300 python3 tools/span_loop_runs.py $D/cig 12 --unphased - 2>&1 | cut -c1-150 | tee -a $OUT/cig_runs.txt
echo "--- 1.5 s between runs" | tee -a $OUT/cig_runs.txt
PAUSE_S=1.5 timeout -k 10 300 python3 tools/span_loop_runs.py $D/cig 12 --unphased - 2>&1 | cut -c1-150 | tee -a $OUT/cig_runs.txt
echo "--- back to back again" | tee -a $OUT/cig_runs.txt
timeout -k 10 300 python3 tools/span_loop_runs.py $D/cig 12 --unphased - 2>&1 | cut -c1-150 | tee -a $OUT/cig_runs.txt
rm -rf $D
