#!/bin/bash
# GPU box, round 3: copies through shader kernels instead of the SDMA engines (HSA_ENABLE_SDMA=0: the first copy of a process costs
# 0.3 instead of 8 - 19 ms) - what do uploads and the whole call look like?
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03sd
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
CLI=$ROOT/inquistr_amd/lib/inquistr
python3 tools/make_synth_bam.py unphased100k 100000 /tmp/sd native > $OUT/gen.log 2>&1 || { tail $OUT/gen.log; exit 1; }
INQ_FRONTEND=device $CLI call /tmp/sd.bam -R /tmp/sd.bed -t 16 -u --sample-name S > /tmp/sd_ref.inq
for round in 1 2; do for mode in 1 0; do
  for rep in 1 2 3 4 5; do
    s=$(date +%s%N); HSA_ENABLE_SDMA=$mode INQ_FRONTEND=device $CLI call /tmp/sd.bam -R /tmp/sd.bed -t 16 -u --sample-name S > /tmp/sd_o.inq; e=$(date +%s%N)
    echo "sdma=$mode run $rep: $(( (e - s) / 1000000 )) ms $(cmp -s /tmp/sd_o.inq /tmp/sd_ref.inq && echo same || echo DIFFERENT)" | tee -a $OUT/sdma_ab.txt
  done
  HSA_ENABLE_SDMA=$mode INQ_TIMING=2 INQ_FRONTEND=device $CLI call /tmp/sd.bam -R /tmp/sd.bed -t 16 -u --sample-name S 2> $OUT/trace_$mode.err > /dev/null
  echo "sdma=$mode uploads: $(grep 'upload ' $OUT/trace_$mode.err | sed 's/.*upload \([0-9.]*\) ms.*/\1/' | tr '\n' ' ') | $(grep 'device front end:' $OUT/trace_$mode.err | sed 's/.*spans/spans/')" | tee -a $OUT/sdma_ab.txt
done; done
