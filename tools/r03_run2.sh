#!/bin/bash
# GPU box, round 3, second call: new front-end tests (deferred spans, sessions), pin probe, start-up split, cohort timing.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03b
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_front.py tests/test_gpu_end_to_end.py tests/test_abi_and_host.py -m gpu -x -q > $OUT/gputest.log 2>&1; echo "pytest rc $?" | tee -a $OUT/gputest.log; tail -15 $OUT/gputest.log
$ROOT/inquistr_amd/lib/pin_probe > $OUT/pin_probe.txt 2>&1; cat $OUT/pin_probe.txt
CLI=$ROOT/inquistr_amd/lib/inquistr
python3 tools/make_synth_bam.py unphased100k 100000 /tmp/c0 native > $OUT/gen.txt 2>&1; cat $OUT/gen.txt
for k in 1 2 3 4 5 6 7; do cp /tmp/c0.bam /tmp/c$k.bam; cp /tmp/c0.bam.bai /tmp/c$k.bam.bai; done
mkdir -p /tmp/coh /tmp/sepout
for i in 1 2 3; do
  t0=$(date +%s.%N); INQ_FRONTEND=device INQ_TIMING=2 $CLI call /tmp/c0.bam -R /tmp/c0.bed -t 16 -u > /tmp/sepout/c0.inq 2> $OUT/single_run$i.err; t1=$(date +%s.%N)
  python3 -c "print('single 1GB run $i: process wall %.3f s' % ($t1 - $t0))" | tee -a $OUT/cohort_walls.txt
done
grep "inq ctx\]" $OUT/single_run1.err $OUT/single_run2.err $OUT/single_run3.err | tee $OUT/ctx_split.txt
for i in 1 2 3 4 5; do
  t0=$(date +%s.%N); INQ_FRONTEND=device INQ_TIMING=2 $CLI cohort -R /tmp/c0.bed -t 16 -u --out-dir /tmp/coh /tmp/c0.bam /tmp/c1.bam /tmp/c2.bam /tmp/c3.bam /tmp/c4.bam /tmp/c5.bam /tmp/c6.bam /tmp/c7.bam 2> $OUT/cohort_run$i.err; t1=$(date +%s.%N)
  python3 -c "print('cohort 8 x 1GB run $i: process wall %.3f s' % ($t1 - $t0))" | tee -a $OUT/cohort_walls.txt
  grep "inq session\]" $OUT/cohort_run$i.err | cut -c1-160 | tee -a $OUT/cohort_walls.txt
done
for k in 0 1 2 3 4 5 6 7; do cmp /tmp/coh/c$k.inq /tmp/sepout/c0.inq || echo "DIFF c$k"; done; echo "cohort outputs compared" | tee -a $OUT/cohort_walls.txt
# the 4 GB CIGAR-only file again: locus kernels per launch now
python3 tools/make_synth_bam.py unphased100k 400000 /tmp/big native > /dev/null 2>&1
for i in 1 2 3; do
  t0=$(date +%s.%N); INQ_FRONTEND=device INQ_TIMING=2 $CLI call /tmp/big.bam -R /tmp/big.bed -t 16 -u --sample-name S > /tmp/big.inq 2> $OUT/l2_4GB_run$i.err; t1=$(date +%s.%N)
  python3 -c "print('4GB run $i: process wall %.3f s' % ($t1 - $t0))" | tee -a $OUT/l2_4GB_walls.txt
  grep "timing\] device\|inq call\]" $OUT/l2_4GB_run$i.err | cut -c1-260 | tee -a $OUT/l2_4GB_walls.txt
done
cd /tmp && INQ_FAST_EXIT=0 INQ_FRONTEND=device rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cli4GB_trace -- $CLI call /tmp/big.bam -R /tmp/big.bed -t 16 -u --sample-name S > /tmp/big_prof.inq 2> $OUT/cli4GB_trace.log; echo "rocprof rc $?"
cd $ROOT
cmp /tmp/big.inq /tmp/big_prof.inq && echo "profiled output identical"
for f in $(find $OUT/cli4GB_trace -name '*kernel_stats.csv'); do cp $f $OUT/cli4GB_kernel_stats.csv; done
f=$(find $OUT/cli4GB_trace -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && (head -1 $f; grep -E 'locus_call' $f) > $OUT/cli4GB_locus_call_trace.csv
rm -rf $OUT/cli4GB_trace
ls $OUT
