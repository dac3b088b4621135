#!/bin/bash
# GPU box, round 4: the span loop after (a) outgrown buffers are retired instead of freed, (b) inflate_ahead on, (c) pooled reader
# threads spread over L3 domains, (d) four slots per file.  GPU tests first (the defaults changed), then runs of the 12.8 GB and 1.0 GB files.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04_loop2
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests.txt 2>&1; echo "gpu tests rc $?"; tail -4 $OUT/gpu_tests.txt
D=/tmp/inq_r04; mkdir -p $D
( time timeout -k 10 300 python3 tools/make_synth_bam.py unphased100k 100000 $D/cig native 1 ) 2>&1 | grep -E "wrote|real"
( time timeout -k 10 600 python3 tools/make_synth_bam.py unphased100k ${1:-40000} $D/seq native-seq 1 ) 2>&1 | grep -E "wrote|real"
timeout -k 10 300 python3 tools/span_loop_runs.py $D/seq 12 --unphased - INQ_IO_PIN=0 INQ_INFLATE_AHEAD=0 2>&1 | tee $OUT/seq_runs.txt
timeout -k 10 200 python3 tools/span_loop_runs.py $D/cig 8 --unphased - INQ_INFLATE_AHEAD=0 2>&1 | tee $OUT/cig_runs.txt
INQ_FRONTEND=device INQ_TIMING=2 timeout -k 10 120 inquistr_amd/lib/inquistr call $D/seq.bam -R $D/seq.bed -t 16 -u --sample-name S > /dev/null 2> $OUT/seq_trace.err
INQ_FRONTEND=device INQ_TIMING=2 timeout -k 10 120 inquistr_amd/lib/inquistr call $D/cig.bam -R $D/cig.bed -t 16 -u --sample-name S > /dev/null 2> $OUT/cig_trace.err
rm -rf $D
