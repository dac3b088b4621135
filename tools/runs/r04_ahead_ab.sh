#!/bin/bash
# GPU box, round 4: inflate_ahead off / on for one-file processes - where did round 3's +0.3 s come from?
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
sync
for f in cig seq; do
  for ahead in 0 1 0 1; do
    for i in 0 1 2; do
      s=$(date +%s.%N)
      INQ_INFLATE_AHEAD=$ahead INQ_FRONTEND=device INQ_TIMING=1 timeout -k 10 120 $CLI call $D/$f.bam -R $D/$f.bed -t 16 -u --sample-name S > $D/o_${f}_$ahead.inq 2> $OUT/${f}_ahead${ahead}_$i.err || echo "run failed"
      e=$(date +%s.%N)
      echo "$f ahead=$ahead run $i wall $(echo "$e - $s" | bc -l) s :: $(grep 'device front end' $OUT/${f}_ahead${ahead}_$i.err | cut -c14-)" | tee -a $OUT/summary.txt
    done
  done
  INQ_INFLATE_AHEAD=1 INQ_FRONTEND=device INQ_TIMING=2 timeout -k 10 120 $CLI call $D/$f.bam -R $D/$f.bed -t 16 -u --sample-name S > /dev/null 2> $OUT/${f}_ahead1_trace.err
  INQ_INFLATE_AHEAD=0 INQ_FRONTEND=device INQ_TIMING=2 timeout -k 10 120 $CLI call $D/$f.bam -R $D/$f.bed -t 16 -u --sample-name S > /dev/null 2> $OUT/${f}_ahead0_trace.err
  cmp $D/o_${f}_0.inq $D/o_${f}_1.inq && echo "$f: outputs identical"
done
rm -rf $D
