#!/bin/bash
# GPU box: stage timings of the CLI on a BAM with real-record-shaped reads.  Usage: tools/e2e_seq_timing.sh [loci]
LOCI=${1:-3000}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
python3 $ROOT/tools/make_synth_bam.py unphased100k $LOCI /tmp/e2e_s seq > /dev/null || exit 1
CLI="$ROOT/inquistr_amd/lib/inquistr call /tmp/e2e_s.bam -R /tmp/e2e_s.bed -t 16 -u --sample-name S"
for i in 1 2 3 4; do
  t0=$(date +%s.%N); INQ_FRONTEND=device INQ_TIMING=2 $CLI > /tmp/e2e_s.inq 2> /tmp/e2e_s.err; t1=$(date +%s.%N)
  python3 -c "print('device front end: process wall %.3f s' % ($t1 - $t0))"
  grep "inq loader\|inq span\]\|timing\] device" /tmp/e2e_s.err | cut -c1-230
done
