#!/bin/bash
# GPU box, round 4: (1) is the slow run in five the cgroup's CPU quota?  cpu.stat (nr_throttled, throttled_usec) around every run, at
# 16 / 12 / 8 reader threads;  (2) inflate_ahead off / on for one-file processes - where did round 3's +0.3 s come from?
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04_ahead
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
D=/tmp/inq_r04; mkdir -p $D
CLI=inquistr_amd/lib/inquistr
( time timeout -k 10 300 python3 tools/make_synth_bam.py unphased100k 100000 $D/cig native 1 ) 2>&1 | grep -E "wrote|real"
( time timeout -k 10 600 python3 tools/make_synth_bam.py unphased100k ${1:-40000} $D/seq native-seq 1 ) 2>&1 | grep -E "wrote|real"
stat() { grep -E "nr_throttled|throttled_usec|usage_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; }
one() {  # file threads ahead tag
  local b=$(stat) s=$(date +%s.%N)
  INQ_INFLATE_AHEAD=$3 INQ_FRONTEND=device INQ_TIMING=1 timeout -k 10 120 $CLI call $D/$1.bam -R $D/$1.bed -t $2 -u --sample-name S > $D/o_$1_$3.inq 2> $OUT/$4.err || echo "run failed"
  local e=$(date +%s.%N) a=$(stat)
  echo "$1 -t $2 ahead=$3 wall $(echo "$e - $s" | bc -l | cut -c1-6) s :: $(grep 'span loop' $OUT/$4.err | sed -E 's/.*compressed, //') :: $(grep 'device front end' $OUT/$4.err | sed -E 's/.*spans /spans /') :: before [$b] after [$a]" | tee -a $OUT/summary.txt
}
echo "--- throttling? 10 runs at -t 16" | tee -a $OUT/summary.txt
for i in 0 1 2 3 4 5 6 7 8 9; do one seq 16 0 thr16_$i; done
for t in 12 8; do
  echo "--- -t $t" | tee -a $OUT/summary.txt
  for i in 0 1 2 3 4; do one seq $t 0 thr${t}_$i; done
done
for f in cig seq; do
  echo "--- inflate_ahead A/B on $f" | tee -a $OUT/summary.txt
  for ahead in 0 1 0 1; do
    for i in 0 1 2; do one $f 16 $ahead ${f}_ahead${ahead}_$i; done
  done
  INQ_INFLATE_AHEAD=1 INQ_FRONTEND=device INQ_TIMING=2 timeout -k 10 120 $CLI call $D/$f.bam -R $D/$f.bed -t 16 -u --sample-name S > /dev/null 2> $OUT/${f}_ahead1_trace.err
  INQ_INFLATE_AHEAD=0 INQ_FRONTEND=device INQ_TIMING=2 timeout -k 10 120 $CLI call $D/$f.bam -R $D/$f.bed -t 16 -u --sample-name S > /dev/null 2> $OUT/${f}_ahead0_trace.err
  cmp $D/o_${f}_0.inq $D/o_${f}_1.inq && echo "$f: outputs identical"
done
rm -rf $D
