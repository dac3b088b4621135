#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04_cigslow
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
D=/tmp/inq_r04; mkdir -p $D
( time timeout -k 10 300 python3 tools/make_synth_bam.py unphased100k 100000 $D/cig native 6 ) 2>&1 | grep -E "wrote|real"
timeout -k 10 300 python3 tools/span_loop_runs.py $D/cig 16 --unphased --keep-slow $OUT/err - 2>&1 | tee $OUT/cig_runs.txt
rm -rf $D
