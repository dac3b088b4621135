#!/bin/bash
# GPU box: why is the FIRST launch of the locus kernels in the CLI (cold behind the gather) slower than the same launch repeated?
# rocprofv3 --pmc passes on the CLI itself (INQ_CALL_AGAIN=1: every flush launches twice), counters of locus_call_small per dispatch.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_cli_locus
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/tools/make_synth_bam.py unphased100k 100000 /tmp/cli_prof native 6 > $OUT/gen.log 2>&1 || { tail $OUT/gen.log; exit 1; }
# INQ_CALL_AGAIN is read by a measurement build only: make -C inquistr_amd/csrc -B DEBUG_ENV=1 (and rebuild without it afterwards)
export INQ_FRONTEND=device INQ_FAST_EXIT=0 INQ_CALL_AGAIN=1
CLI="$ROOT/inquistr_amd/lib/inquistr call /tmp/cli_prof.bam -R /tmp/cli_prof.bed -t 16 -u --sample-name S --ctx-option inflate_ahead=0"
i=0
# (one set only: the TCP_UTCL1_* / TCC_EA0_* sets made rocprofv3 abort and the run hang on this pool - round 4 lost eight GPU-minutes to it)
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- $CLI > /dev/null 2> $OUT/p$i.log || { tail -5 $OUT/p$i.log; }
  f=$(find $OUT/p$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && (head -1 $f; grep -E 'locus_call_small' $f) > $OUT/pmc_$i.csv
  rm -rf $OUT/p$i
done
timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- $CLI > /dev/null 2> $OUT/t.log
f=$(find $OUT/t -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && (head -1 $f; grep -E 'locus_call_small|cigar_gather|join_kernel' $f) > $OUT/kernel_trace.csv
rm -rf $OUT/t /tmp/cli_prof.*
python3 - <<'PY'
import csv, os, collections
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/prof_cli_locus")
for i in (1, 2, 3):
    p = os.path.join(out, f"pmc_{i}.csv")
    if not os.path.exists(p): continue
    rows = list(csv.DictReader(open(p)))
    by = collections.OrderedDict()
    for r in rows:
        by.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for d, c in by.items():
        print("dispatch", d, {k: round(v) for k, v in c.items()})
p = os.path.join(out, "kernel_trace.csv")
if os.path.exists(p):
    for r in csv.DictReader(open(p)):
        if "locus_call_small" in r["Kernel_Name"]:
            print("trace", r["Dispatch_Id"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us")
PY
