#!/bin/bash
# GPU box, round 3: where a block's time goes (clock probe build), one token list against two streams.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03x
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
for k in "cigar 6" "cigar 1" "ont 6"; do set -- $k
  for lib in base new; do for tk in 0 1; do
    echo "== $1 level $2 $lib tokens=$tk" | tee -a $OUT/inflate_phase_probe.txt
    INQ_INFLATE_DEBUG=8 INQ_LIB=$ROOT/inquistr_amd/lib/libinq_dbg_$lib.so TOKENS=$tk ALGO=0 timeout -k 10 200 python3 tools/inflate_bench.py 20000 $2 $1 2>&1 | grep -v amdgpu.ids | tail -12 | sed 's/comp .* kernel/kernel/' | tee -a $OUT/inflate_phase_probe.txt
  done; done
done
