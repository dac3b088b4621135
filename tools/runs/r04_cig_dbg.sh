#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04_cigdbg
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
D=/tmp/inq_r04; mkdir -p $D
( time timeout -k 10 300 python3 tools/make_synth_bam.py unphased100k 100000 $D/cig native 1 ) 2>&1 | grep -E "wrote|real"
timeout -k 10 300 python3 tools/span_loop_runs.py $D/cig 6 --unphased - INQ_INFLATE_AHEAD=0 INQ_SPAN_BUFFERS=3 INQ_SPAN_BUFFERS=3,INQ_INFLATE_AHEAD=0 INQ_IO_PIN=0 INQ_IO_PIN=0,INQ_INFLATE_AHEAD=0,INQ_SPAN_BUFFERS=3 2>&1 | tee $OUT/cig_runs.txt
for v in 1 0; do
INQ_INFLATE_AHEAD=$v INQ_FRONTEND=device INQ_TIMING=2 timeout -k 10 120 inquistr_amd/lib/inquistr call $D/cig.bam -R $D/cig.bed -t 16 -u --sample-name S > /dev/null 2> $OUT/cig_trace_ahead$v.err
INQ_INFLATE_AHEAD=$v INQ_FRONTEND=device INQ_TIMING=2 timeout -k 10 120 inquistr_amd/lib/inquistr call $D/cig.bam -R $D/cig.bed -t 16 -u --sample-name S > /dev/null 2> $OUT/cig_trace_ahead${v}_b.err
done
rm -rf $D
