#!/bin/bash
# GPU box, round 3: spans inflated right behind their upload (inq_span_stage, own stream) - GPU suite, then A/B of the span loop
# and of the served throughput (INQ_INFLATE_AHEAD=0/1).
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03ah
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $OUT/gputest_full.log 2>&1; rc=$?; echo "pytest rc $rc" | tee -a $OUT/gputest_full.log; tail -3 $OUT/gputest_full.log
[ $rc = 0 ] || exit 1
CLI=$ROOT/inquistr_amd/lib/inquistr
python3 tools/make_synth_bam.py unphased100k 100000 /tmp/ah native > $OUT/gen.log 2>&1 || { tail $OUT/gen.log; exit 1; }
python3 tools/make_synth_bam.py unphased100k 12000 /tmp/ahs native-seq > $OUT/gen2.log 2>&1 || { tail $OUT/gen2.log; exit 1; }
ls -la /tmp/ah.bam /tmp/ahs.bam | tee $OUT/sizes.txt
for mode in 0 1 0 1; do
  for f in ah ahs; do
    INQ_INFLATE_AHEAD=$mode INQ_TIMING=2 INQ_FRONTEND=device $CLI serve --socket /tmp/ah.sock --idle-exit 60 2> $OUT/server_${mode}_$f.err &
    SP=$!
    for i in $(seq 1 100); do [ -S /tmp/ah.sock ] && break; sleep 0.05; done
    INQ_SERVER=/tmp/ah.sock $CLI call /tmp/$f.bam -R /tmp/$f.bed -t 16 -u --sample-name S > /tmp/ah_warm.inq
    s=$(date +%s%N)
    for i in 1 2 3 4 5 6 7 8; do INQ_SERVER=/tmp/ah.sock $CLI call /tmp/$f.bam -R /tmp/$f.bed -t 16 -u --sample-name S > /tmp/ah_$i.inq & done
    wait $(jobs -p | grep -v "^$SP$") 2>/dev/null
    e=$(date +%s%N)
    echo "ahead=$mode $f: 8 callers at once $(( (e - s) / 1000000 )) ms; $(grep -c 'inq timing. device front end' $OUT/server_${mode}_$f.err) calls; last: $(grep 'device front end:' $OUT/server_${mode}_$f.err | tail -1 | sed 's/.*spans/spans/')" | tee -a $OUT/ahead_ab.txt
    cmp /tmp/ah_1.inq /tmp/ah_warm.inq || echo "DIFFERENT OUTPUT" | tee -a $OUT/ahead_ab.txt
    $CLI serve --socket /tmp/ah.sock --quit
    wait $SP
  done
done
