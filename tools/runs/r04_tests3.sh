#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04_tests3
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q -s -k "million or four_ranks or two_ranks or 100000" > $OUT/gpu_tests_new.txt 2>&1; echo "new gpu tests rc $?"; grep -E "launch sequence|passed|failed|Error|error" $OUT/gpu_tests_new.txt | tail -12
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests.txt 2>&1; echo "gpu tests rc $?"; tail -4 $OUT/gpu_tests.txt
D=/tmp/inq_r04; mkdir -p $D
( time timeout -k 10 300 python3 tools/make_synth_bam.py unphased100k 100000 $D/cig native 6 ) 2>&1 | grep -E "wrote|real"
SHOW_CALLS=1 timeout -k 10 300 python3 tools/span_loop_runs.py $D/cig 5 --unphased - INQ_INFLATE_AHEAD=0 INQ_FLUSH_LOCI=100000 2>&1 | tee $OUT/locus_runs.txt
rm -rf $D
