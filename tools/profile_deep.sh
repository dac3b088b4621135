#!/bin/bash
# GPU box: kernel trace of the launch sequence on a batch with a 1 000 000-read locus (tests/test_gpu_parity.py -k million): how long
# locus_call_mid_walk (the walk, dealt over the grid) and the persistent locus_call_tail (phase A + the grid-wide selects between
# barriers) take.  -> gpurun_out/prof_deep/kernel_stats_deep.csv
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_deep
mkdir -p $OUT
cd $ROOT && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k million > $OUT/trace.log 2>&1 || { tail -20 $OUT/trace.log; exit 1; }
f=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
(head -1 $f; grep -E 'locus_call' $f) > $OUT/kernel_trace_deep.csv
for f in $(find $OUT/trace -name '*kernel_stats.csv'); do (head -1 $f; grep -E 'locus_call' $f) > $OUT/kernel_stats_deep.csv; done
rm -rf $OUT/trace
grep -E "launch sequence|passed" $OUT/trace.log
cat $OUT/kernel_stats_deep.csv
