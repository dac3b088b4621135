#!/bin/bash
# GPU box: SQ counters of the inflate kernel on the microbench.  Usage: tools/profile_inflate.sh <blocks> <level> <kind>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_inflate_$3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  n=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -- python3 $ROOT/tools/inflate_bench.py $BLOCKS $LEVEL $KIND > $OUT/$n.log 2>&1 || tail -5 $OUT/$n.log
  f=$(find $OUT/$n -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if "bgzf_inflate" in r["Kernel_Name"]:
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(agg): print(f"{k:28s} {agg[k] / max(n[k], 1):16.0f}  (mean of {n[k]} dispatches)")
PY
  rm -rf $OUT/$n
}
BLOCKS=$1; LEVEL=$2; KIND=$3
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run b SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
run c SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC
