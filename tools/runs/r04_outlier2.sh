#!/bin/bash
# GPU box, round 4: the slow SECOND read of a freshly written file: /proc/vmstat's pgactivate beside every run, without and with
# POSIX_FADV_NOREUSE on the loader's descriptor (two files, so that each variant meets a file in the same state).
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04_outlier2
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
uname -r | tee $OUT/kernel.txt
D=/tmp/inq_r04; mkdir -p $D
( time timeout -k 10 600 python3 tools/make_synth_bam.py unphased100k ${1:-40000} $D/seq native-seq 1 ) 2>&1 | grep -E "wrote|real"
timeout -k 10 300 python3 tools/span_loop_runs.py $D/seq 6 --unphased INQ_FADV_NOREUSE=0 2>&1 | tee $OUT/runs_without_noreuse.txt
rm -f $D/seq.bam $D/seq.bam.bai
( time timeout -k 10 600 python3 tools/make_synth_bam.py unphased100k ${1:-40000} $D/seq native-seq 1 ) 2>&1 | grep -E "wrote|real"
timeout -k 10 300 python3 tools/span_loop_runs.py $D/seq 12 --unphased - 2>&1 | tee $OUT/runs_with_noreuse.txt
rm -rf $D
