#!/bin/bash
# Span loop rate of ONE rank as a function of its reader threads: the CLI on a SEQ-bearing file with LOCAL_WORLD_SIZE = 1, 2, 4, 8
# (the reader pool takes granted cores / LOCAL_WORLD_SIZE threads, at least 2: host/span_pipeline.cc span_io_threads).
# usage (GPU box): bash tools/reader_threads_sweep.sh [loci] [runs]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
LOCI=${1:-24000}; RUNS=${2:-3}
D=/tmp/inq_rts; mkdir -p $D
[ -f $D/f.bam ] || timeout -k 10 400 python3 tools/make_synth_bam.py unphased100k $LOCI $D/f native-seq 6 | tail -1
cat $D/f.bam > /dev/null; cat $D/f.bam > /dev/null
for lws in 1 2 4 8; do
  for r in $(seq $RUNS); do
    sleep 1.2
    LOCAL_WORLD_SIZE=$lws LOCAL_RANK=0 INQ_FRONTEND=device INQ_TIMING=1 timeout -k 10 120 inquistr_amd/lib/inquistr call $D/f.bam -R $D/f.bed -t 16 -u --sample-name S 2> $D/err > $D/out.inq
    echo "LOCAL_WORLD_SIZE=$lws run $r: $(grep -o 'span loop:.*' $D/err) | $(grep -o 'waiting for the loader [0-9.]*s' $D/err)"
  done
done
