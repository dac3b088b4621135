#!/bin/bash
# Many BAMs on one context: `inquistr cohort` over K copies-by-seed of a SEQ-bearing file (each its own bytes in the page cache) -
# whole command seconds, aggregate GB/s of BAM, against K single `inquistr call`s.  usage (GPU box): bash tools/cohort_throughput.sh [loci] [K]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
LOCI=${1:-6000}; K=${2:-8}
D=/tmp/inq_cohort; mkdir -p $D $D/out
timeout -k 10 300 python3 tools/make_synth_bam.py unphased100k $LOCI $D/s0 native-seq 6 | tail -1
for k in $(seq 1 $((K-1))); do cp $D/s0.bam $D/s$k.bam; cp $D/s0.bam.bai $D/s$k.bam.bai 2>/dev/null || cp $D/s0.bai $D/s$k.bai; done
for k in $(seq 0 $((K-1))); do cat $D/s$k.bam > /dev/null; cat $D/s$k.bam > /dev/null; done
BYTES=$(stat -c %s $D/s0.bam)
FILES=$(for k in $(seq 0 $((K-1))); do echo -n "$D/s$k.bam "; done)
for r in 1 2 3; do
  sleep 1.2
  s=$(date +%s.%N)
  INQ_FRONTEND=device INQ_TIMING=1 timeout -k 10 300 inquistr_amd/lib/inquistr cohort -R $D/s0.bed -t 16 -u --out-dir $D/out $FILES 2> $D/err
  e=$(date +%s.%N)
  python3 -c "
dt=$e-$s; b=$BYTES*$K
print(f'cohort run $r: $K files x {$BYTES/1e9:.2f} GB in {dt:.3f} s whole command = {b/1e9/dt:.1f} GB/s of BAM ({dt/$K*1e3:.1f} ms per file)')"
  grep -c "span loop" $D/err | sed 's/^/  span loops: /'
done
s=$(date +%s.%N)
for k in $(seq 0 $((K-1))); do INQ_FRONTEND=device timeout -k 10 60 inquistr_amd/lib/inquistr call $D/s$k.bam -R $D/s0.bed -t 16 -u > $D/out/single_$k.inq; done
e=$(date +%s.%N)
python3 -c "print(f'$K single calls back to back: {$e-$s:.3f} s')"
cmp $D/out/single_0.inq $D/out/s0.inq && echo "cohort rows = single call rows"
