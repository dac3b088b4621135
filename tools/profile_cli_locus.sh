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
# Round 4 lost eight GPU-minutes here.  What happened (gpurun_out/prof_cli_locus/p2.log of that run): the counter set of the second pass
# (TCP_UTCL1_* + TCC_EA0_*: two blocks, more counters than ONE pass has slots for) made rocprofiler_create_counter_config fail with
# "error code 38: Request exceeds the capabilities of the hardware to collect"; the tool library reports that through glog FATAL =
# abort(), raised ON THE THREAD THAT MADE THE FIRST HIP CALL - the CLI's context thread, inside inq_ctx_create_early; rocprofv3's signal
# handler then "finalized" on that thread and never came back, and the CLI's other threads waited for a context that would never be
# published (uploader: polling stage_ready; caller: SpanPipeline::next(); a join of the context thread) - no wait had a bound, so the
# process stayed until gpurun's limit.  Since round 5 every wait for the context thread is bounded (host/driver_internal.h AsyncCtx:
# INQ_CTX_TIMEOUT_S, exit status 1), and this script (a) checks every counter name against what the device offers BEFORE a pass,
# (b) keeps a pass to one block's counters, at most four, (c) stops at the first pass that fails.
AVAIL=$OUT/counters_offered.txt  # (a file, not a pipe: grep -q ends a pipe early and pipefail would call that a failure)
(rocprofv3 -L || rocprofv3-avail list) > $AVAIL 2>&1
grep -q "Counter_Name.*SQ_WAVES" $AVAIL || { echo "cannot list the device's counters (rocprofv3 -L / rocprofv3-avail list): no pass is run"; head -5 $AVAIL; exit 1; }
check_set() {  # every name must be offered; a pass holds counters of ONE block (prefix up to the first '_'), four at most
  local n=0 blk=""
  for c in $1; do
    grep -qE "Counter_Name[[:space:]]*:[[:space:]]*$c\$" $AVAIL || { echo "counter $c is not offered by this device: pass skipped"; return 1; }
    [ -z "$blk" ] && blk=${c%%_*}
    [ "${c%%_*}" = "$blk" ] || { echo "counters of two blocks in one pass ($blk, ${c%%_*}): pass skipped"; return 1; }
    n=$((n+1))
  done
  [ $n -le 4 ] || { echo "$n counters in one pass (at most 4): pass skipped"; return 1; }
}
export INQ_CTX_TIMEOUT_S=30
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  check_set "$set" || continue
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- $CLI > /dev/null 2> $OUT/p$i.log
  rc=$?
  if [ $rc -ne 0 ]; then echo "pass $i ($set) failed with status $rc: stopping"; tail -5 $OUT/p$i.log; exit 1; fi
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
for i in (1, 2):
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
