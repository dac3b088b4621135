#!/bin/bash
# GPU box: phase timings of the CLI on a synthetic BAM.  Usage: tools/e2e_timing.sh <loci> [threads]
set -o pipefail
LOCI=${1:-100000}; THREADS=${2:-16}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
python3 $ROOT/tools/make_synth_bam.py unphased100k $LOCI /tmp/e2e_t || exit 1
CLI="$ROOT/inquistr_amd/lib/inquistr call /tmp/e2e_t.bam -R /tmp/e2e_t.bed -t $THREADS -u --sample-name S"
for fe in device device device host; do
  t0=$(date +%s.%N)
  INQ_FRONTEND=$fe INQ_TIMING=1 $CLI > /tmp/e2e_$fe.inq
  t1=$(date +%s.%N)
  python3 -c "print('$fe front end: process wall %.3f s' % ($t1 - $t0))"
done
cmp /tmp/e2e_device.inq /tmp/e2e_host.inq && echo "outputs identical"
