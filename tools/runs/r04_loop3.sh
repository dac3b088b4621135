#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04_loop3
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout -k 10 60 inquistr_amd/lib/stream_probe 4 2>&1 | grep -E "more streams|per turn" | tee $OUT/stream_probe_parallel.txt
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests.txt 2>&1; echo "gpu tests rc $?"; tail -4 $OUT/gpu_tests.txt
D=/tmp/inq_r04; mkdir -p $D
( time timeout -k 10 600 python3 tools/make_synth_bam.py unphased100k ${1:-40000} $D/seq native-seq 6 ) 2>&1 | grep -E "wrote|real"
timeout -k 10 300 python3 tools/span_loop_runs.py $D/seq 14 --unphased - 2>&1 | tee $OUT/seq_runs.txt
INQ_FRONTEND=device INQ_TIMING=2 timeout -k 10 120 inquistr_amd/lib/inquistr call $D/seq.bam -R $D/seq.bed -t 16 -u --sample-name S > /dev/null 2> $OUT/seq_trace.err
( time timeout -k 10 300 python3 tools/make_synth_bam.py unphased100k 100000 $D/cig native 6 ) 2>&1 | grep -E "wrote|real"
timeout -k 10 300 python3 tools/span_loop_runs.py $D/cig 8 --unphased - 2>&1 | tee $OUT/cig_runs.txt
rm -rf $D
