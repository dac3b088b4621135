#!/bin/bash
# GPU box, round 3, third call: loader gate A/B (runtime start-up vs early reads), exit modes, SEQ-bearing file, bench line.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03c
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
CLI=$ROOT/inquistr_amd/lib/inquistr
python3 tools/make_synth_bam.py unphased100k 100000 /tmp/c0 native > $OUT/gen.txt 2>&1
python3 tools/make_synth_bam.py unphased100k 400000 /tmp/big native >> $OUT/gen.txt 2>&1
python3 tools/make_synth_bam.py unphased100k 15000 /tmp/seq native-seq >> $OUT/gen.txt 2>&1; cat $OUT/gen.txt
run() { # name file extra-env...
  local name=$1 f=$2; shift 2
  for i in 1 2 3 4 5; do
    t0=$(date +%s.%N); env "$@" INQ_FRONTEND=device INQ_TIMING=2 $CLI call /tmp/$f.bam -R /tmp/$f.bed -t 16 -u --sample-name S > /tmp/$f.$name.inq 2> $OUT/${f}_${name}_run$i.err; t1=$(date +%s.%N)
    python3 -c "print('$f $name run $i: process wall %.3f s' % ($t1 - $t0))" | tee -a $OUT/walls.txt
    grep "timing\] device" $OUT/${f}_${name}_run$i.err | cut -c1-230 | tee -a $OUT/walls.txt
    grep "inq ctx\] @" $OUT/${f}_${name}_run$i.err | cut -c1-100 | tee -a $OUT/walls.txt
  done
}
for f in c0 seq big; do
  run gated $f INQ_X=1
  run early $f INQ_EARLY_READ=1
  run gated_fullexit $f INQ_FAST_EXIT=0
done
cmp /tmp/big.gated.inq /tmp/big.early.inq && cmp /tmp/seq.gated.inq /tmp/seq.gated_fullexit.inq && echo "outputs identical" | tee -a $OUT/walls.txt
for thr in 16; do
  t0=$(date +%s.%N); $ROOT/oracle/ref_shaped_call /tmp/seq.bam /tmp/seq.bed B $thr 1 5 3 S > /tmp/seq_B.inq; t1=$(date +%s.%N)
  python3 -c "print('seq CPU B $thr threads: %.3f s' % ($t1 - $t0))" | tee -a $OUT/walls.txt
done
cmp /tmp/seq.gated.inq /tmp/seq_B.inq && echo "seq == CPU B" | tee -a $OUT/walls.txt
rm -f /tmp/big.bam /tmp/seq.bam /tmp/c0.bam
timeout -k 10 600 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $?"; tail -c 300 $OUT/bench_default.json
ls $OUT | wc -l
