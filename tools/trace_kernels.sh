#!/bin/bash
# kernel-trace stats of one ab_options run (GPU box): tools/trace_kernels.sh <tag> <ab_options args...>
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $ROOT/tools/ab_options.py "$@" > $OUT/log.txt 2>&1
f=$(find $OUT/t -name '*kernel_stats.csv' | head -1)
grep -E 'Name|inq::' $f | cut -d, -f1-4,6,7 | sed 's/void inq:://' 
rm -rf $OUT/t
