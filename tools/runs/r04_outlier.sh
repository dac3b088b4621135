#!/bin/bash
# GPU box, round 4: catch the slow run (one in five to ten) of the 12.8 GB file with the loader's placement stamps on.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04_outlier
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
D=/tmp/inq_r04; mkdir -p $D
( time timeout -k 10 600 python3 tools/make_synth_bam.py unphased100k ${1:-40000} $D/seq native-seq 1 ) 2>&1 | grep -E "wrote|real"
inquistr_amd/lib/pagecache_nodes $D/seq.bam 256 | tee $OUT/pagecache.txt
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag 2>/dev/null | tee $OUT/thp.txt
grep -E "AnonHugePages|HugePages_Total|MemFree" /proc/meminfo | tee -a $OUT/thp.txt
cat /proc/sys/kernel/numa_balancing 2>/dev/null | tee -a $OUT/thp.txt
timeout -k 10 500 python3 tools/span_loop_runs.py $D/seq ${2:-30} --unphased --keep-slow $OUT/err - 2>&1 | tee $OUT/seq_runs.txt
rm -rf $D
