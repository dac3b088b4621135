#!/bin/bash
# GPU box, round 3, first call: the gpu tests, the default bench line, and stage traces of the CLI on large files.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03a
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== host: $(nproc) cpus, affinity $(python3 -c 'import os;print(len(os.sched_getaffinity(0)))'), mem $(free -g | awk '/Mem/{print $2}') GB, tmp $(df -h /tmp | tail -1)" | tee $OUT/host.txt
cat /sys/fs/cgroup/cpu.max 2>/dev/null | tee -a $OUT/host.txt
timeout -k 10 1500 python3 -m pytest tests -m gpu -x -q > $OUT/gputest.log 2>&1; echo "pytest rc $?" | tee -a $OUT/gputest.log; tail -3 $OUT/gputest.log
timeout -k 10 600 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $?"; tail -c 600 $OUT/bench_default.json
# ---- stage traces
CLI=$ROOT/inquistr_amd/lib/inquistr
python3 tools/make_synth_bam.py unphased100k 400000 /tmp/big native > $OUT/gen_big.txt 2>&1; cat $OUT/gen_big.txt
for i in 1 2 3 4 5; do
  t0=$(date +%s.%N); INQ_FRONTEND=device INQ_TIMING=2 $CLI call /tmp/big.bam -R /tmp/big.bed -t 16 -u --sample-name S > /tmp/big.inq 2> $OUT/l2_4GB_run$i.err; t1=$(date +%s.%N)
  python3 -c "print('4GB run $i: process wall %.3f s' % ($t1 - $t0))" | tee -a $OUT/l2_4GB_walls.txt
  grep "timing\] device" $OUT/l2_4GB_run$i.err | cut -c1-260 | tee -a $OUT/l2_4GB_walls.txt
done
cd /tmp && INQ_FAST_EXIT=0 INQ_FRONTEND=device rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cli4GB_trace -- $CLI call /tmp/big.bam -R /tmp/big.bed -t 16 -u --sample-name S > /tmp/big_prof.inq 2> $OUT/cli4GB_trace.log; echo "rocprof rc $?"
cd $ROOT
for f in $(find $OUT/cli4GB_trace -name '*kernel_stats.csv'); do cp $f $OUT/cli4GB_kernel_stats.csv; done
f=$(find $OUT/cli4GB_trace -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && (head -1 $f; grep -E 'locus_call' $f) > $OUT/cli4GB_locus_call_trace.csv
rm -rf $OUT/cli4GB_trace
rm -f /tmp/big.bam /tmp/big.bam.bai
python3 tools/make_synth_bam.py unphased100k 15000 /tmp/seq native-seq > $OUT/gen_seq.txt 2>&1; cat $OUT/gen_seq.txt
for i in 1 2 3 4 5; do
  t0=$(date +%s.%N); INQ_FRONTEND=device INQ_TIMING=2 $CLI call /tmp/seq.bam -R /tmp/seq.bed -t 16 -u --sample-name S > /tmp/seq.inq 2> $OUT/l2_seq8GB_run$i.err; t1=$(date +%s.%N)
  python3 -c "print('seq run $i: process wall %.3f s' % ($t1 - $t0))" | tee -a $OUT/l2_seq8GB_walls.txt
  grep "timing\] device" $OUT/l2_seq8GB_run$i.err | cut -c1-260 | tee -a $OUT/l2_seq8GB_walls.txt
done
for thr in 16 $(nproc); do
  t0=$(date +%s.%N); $ROOT/oracle/ref_shaped_call /tmp/seq.bam /tmp/seq.bed B $thr 1 5 3 S > /tmp/seq_B.inq; t1=$(date +%s.%N)
  python3 -c "print('seq CPU B $thr threads: %.3f s' % ($t1 - $t0))" | tee -a $OUT/l2_seq8GB_walls.txt
done
cmp /tmp/seq.inq /tmp/seq_B.inq && echo "seq outputs identical" | tee -a $OUT/l2_seq8GB_walls.txt
ls -la $OUT | head -40
