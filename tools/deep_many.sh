#!/bin/bash
# GPU box: MANY loci of a few hundred to a few thousand reads (a targeted panel at several-hundred-fold depth) through the launch
# sequence: this round's kernels against round 4's (build/r4_ref = git archive of the round-4 head, built here).
# -> gpurun_out/deep_many/result.txt
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/deep_many
mkdir -p $OUT
cd $ROOT
if [ -d build/r4_ref/inquistr_amd/csrc ]; then (cd build/r4_ref/inquistr_amd/csrc && make -j8 OBJDIR=/tmp/r4obj > /tmp/r4_build.log 2>&1 || tail -5 /tmp/r4_build.log); fi
: > $OUT/result.txt
run() {  # label, root, workload, loci, neighbors
  echo -n "$1 $3 $4 loci k=$5: " | tee -a $OUT/result.txt
  INQ_ROOT=$2 timeout -k 10 200 python3 tools/seq_timing.py $3 $4 10 $5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d.get('reads_per_locus', '?'), 'reads/locus: sequence ms', {k: round(v['sequence_ms'], 3) for k, v in d.items() if isinstance(v, dict)}, 'TB/s (hinted)', round(d['hint_again']['frac_of_8TBps_sequence'] * 8, 2))" | tee -a $OUT/result.txt
}
for k in 2 4 8 16 33; do
  run r5 "" phased10k 10000 $k
  [ -f build/r4_ref/inquistr_amd/lib/libinquistr_hip.so ] && run r4 $ROOT/build/r4_ref phased10k 10000 $k
done
