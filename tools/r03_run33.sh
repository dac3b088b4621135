#!/bin/bash
# GPU box, round 3: spans used where they lie in a mapping of the file (INQ_SPAN_MAPPED=1: no read into a buffer, the window page-locked
# for the upload) against the pread loader: same bytes out, loader lines, walls; direct and served, 1.0 GB CIGAR-only and 12.8 GB SEQ-bearing.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03mp
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
CLI=$ROOT/inquistr_amd/lib/inquistr
python3 tools/make_synth_bam.py unphased100k 100000 /tmp/mp native > $OUT/gen.log 2>&1 || { tail $OUT/gen.log; exit 1; }
python3 tools/make_synth_bam.py unphased100k 40000 /tmp/mps native-seq > $OUT/gen2.log 2>&1 || { tail $OUT/gen2.log; exit 1; }
for f in mp mps; do
  INQ_FRONTEND=device $CLI call /tmp/$f.bam -R /tmp/$f.bed -t 16 -u --sample-name S > /tmp/${f}_ref.inq
  for mode in 0 1 0 1; do
    for rep in 1 2 3; do
      s=$(date +%s%N); INQ_SPAN_MAPPED=$mode INQ_FRONTEND=device $CLI call /tmp/$f.bam -R /tmp/$f.bed -t 16 -u --sample-name S > /tmp/${f}_$mode.inq; e=$(date +%s%N)
      echo "$f mapped=$mode direct run $rep: $(( (e - s) / 1000000 )) ms $(cmp -s /tmp/${f}_$mode.inq /tmp/${f}_ref.inq && echo same || echo DIFFERENT)" | tee -a $OUT/mapped_ab.txt
    done
  done
  for mode in 0 1; do
    INQ_SPAN_MAPPED=$mode INQ_TIMING=2 INQ_FRONTEND=device $CLI call /tmp/$f.bam -R /tmp/$f.bed -t 16 -u --sample-name S 2> $OUT/trace_${f}_$mode.err > /dev/null
    echo "$f mapped=$mode: $(grep -c 'mapped file' $OUT/trace_${f}_$mode.err) mapped spans; reads $(grep 'read+tables' $OUT/trace_${f}_$mode.err | sed 's/.*read+tables \([0-9.]*\) ms.*/\1/' | sort -n | awk '{a[NR]=$1} END {print a[int((NR+1)/2)]}') ms median; uploads $(grep 'upload ' $OUT/trace_${f}_$mode.err | sed 's/.*upload \([0-9.]*\) ms.*/\1/' | sort -n | awk '{a[NR]=$1} END {print a[int((NR+1)/2)]}') ms median; $(grep 'device front end:' $OUT/trace_${f}_$mode.err | sed 's/.*spans/spans/')" | tee -a $OUT/mapped_ab.txt
  done
  for mode in 0 1; do
    INQ_SPAN_MAPPED=$mode INQ_FRONTEND=device $CLI serve --socket /tmp/mp.sock --idle-exit 60 2> /dev/null &
    SP=$!
    for i in $(seq 1 100); do [ -S /tmp/mp.sock ] && break; sleep 0.05; done
    INQ_SERVER=/tmp/mp.sock $CLI call /tmp/$f.bam -R /tmp/$f.bed -t 16 -u --sample-name S > /tmp/mp_warm.inq
    s=$(date +%s%N)
    for i in 1 2 3 4 5 6; do INQ_SERVER=/tmp/mp.sock $CLI call /tmp/$f.bam -R /tmp/$f.bed -t 16 -u --sample-name S > /tmp/mp_$i.inq & done
    wait $(jobs -p | grep -v "^$SP$") 2>/dev/null
    e=$(date +%s%N)
    echo "$f mapped=$mode served, 6 callers at once: $(( (e - s) / 1000000 )) ms $(cmp -s /tmp/mp_3.inq /tmp/${f}_ref.inq && echo same || echo DIFFERENT)" | tee -a $OUT/mapped_ab.txt
    $CLI serve --socket /tmp/mp.sock --quit
    wait $SP
  done
done
