#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats + separate PMC passes for bench.py.
# Usage: tools/profile_round.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="$ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-l2 --no-read-peak --no-live-pmc --no-l0-configs $*"
# kernel trace: the driver's own command line (python bench.py, default steps / warmup) minus the legs that start other
# programs (cpu_baseline builds the oracle with make, l2 runs the CLI: no child processes under the profiler), so the average
# duration in kernel_stats.csv is directly comparable with roofline.avg_kernel_ms of the default command
# (--no-l0-configs: the per-config launches of the default line run other workloads through the SAME kernel names; they get traces of their own below)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-l2 --no-read-peak --no-live-pmc --no-l0-configs $* > $OUT/trace.log 2>&1 || { tail -20 $OUT/trace.log; exit 1; }
if [ -z "$*" ]; then  # the other configs' kernel stats (config #2, #5, one shard of #4), same command with --workload
  for wl in phased10k expansion50k shard500k; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$wl -- python3 $ROOT/bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --no-l2 --no-read-peak --no-live-pmc --no-l0-configs > $OUT/trace_$wl.log 2>&1 || { tail -5 $OUT/trace_$wl.log; }
    for f in $(find $OUT/trace_$wl -name '*kernel_stats.csv'); do (head -1 $f; grep -E 'inq::' $f) > $OUT/kernel_stats_$wl.csv; done
    grep '^{' $OUT/trace_$wl.log | tail -1 > $OUT/bench_line_$wl.json
    rm -rf $OUT/trace_$wl
  done
fi
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $BENCH > $OUT/pmc_fetch.log 2>&1 || { tail -20 $OUT/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $BENCH > $OUT/pmc_write.log 2>&1 || { tail -20 $OUT/pmc_write.log; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq -- python3 $BENCH > $OUT/pmc_sq.log 2>&1 || { tail -20 $OUT/pmc_sq.log; }
find $OUT -name '*.csv' | head -40
# keep only the small summaries (kernel_stats + per-dispatch counters of our kernels)
for f in $(find $OUT -name '*kernel_stats.csv'); do cp $f $OUT/kernel_stats.csv; done
for d in pmc_fetch pmc_write pmc_sq; do
  f=$(find $OUT/$d -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && (head -1 $f; grep -E 'locus_call' $f) > $OUT/$d.csv
done
f=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
[ -n "$f" ] && (head -1 $f; grep -E 'locus_call' $f) > $OUT/kernel_trace_locus_call.csv
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
ls -la $OUT
