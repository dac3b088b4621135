#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04_again
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests.txt 2>&1; echo "gpu tests rc $?"; tail -5 $OUT/gpu_tests.txt
D=/tmp/inq_r04; mkdir -p $D
( time timeout -k 10 300 python3 tools/make_synth_bam.py unphased100k 100000 $D/cig native 6 ) 2>&1 | grep -E "wrote|real"
for i in 0 1 2; do
  INQ_CALL_AGAIN=1 INQ_INFLATE_AHEAD=0 INQ_FRONTEND=device INQ_TIMING=2 timeout -k 10 120 inquistr_amd/lib/inquistr call $D/cig.bam -R $D/cig.bed -t 16 -u --sample-name S 2>&1 >/dev/null | grep "inq call" | tee -a $OUT/call_again.txt
done
cd /tmp
INQ_FAST_EXIT=0 INQ_INFLATE_AHEAD=0 INQ_FRONTEND=device timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/prof -o cli -- $ROOT/inquistr_amd/lib/inquistr call $D/cig.bam -R $D/cig.bed -t 16 -u --sample-name S > /dev/null 2> $OUT/prof.err || echo "rocprof failed"
find $OUT/prof -type f | head; for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -14 $f | cut -c1-180; done
rm -rf $D
