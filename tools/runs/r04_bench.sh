#!/bin/bash
# GPU box: the default bench line (what the driver runs), timed.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04_bench
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
s=$(date +%s)
timeout -k 10 ${1:-900} python3 bench.py ${@:2} > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $? in $(( $(date +%s) - s )) s"
tail -c 600 $OUT/bench_default.err
python3 - <<'PY'
import json,os
p=os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),"gpurun_out/r04_bench/bench_default.json")
d=json.loads(open(p).read().strip().splitlines()[-1])
print("value", d["value"], "roofline frac", d["roofline"]["frac"], "frac_of_measured", d["roofline"].get("frac_of_measured"))
print("startup_floor", d.get("startup_floor",{}).get("seconds_median"), "h2d", d.get("h2d_copy_peak",{}).get("value"))
print("l1", {k:d.get("l1",{}).get(k) for k in ("loci_per_s","GBps_host_to_device_incl_kernels","error")})
for k in ("l2","l2_seq","l2_seq_level1","l2_seq_large"):
    b=d.get(k,{})
    if "error" in b or "skipped" in b: print(k, b); continue
    print(k, "loci", b.get("loci"), "bam_mb %.0f"%b.get("bam_mb",0), "level", b.get("zlib_level"), "gen_s %.1f"%b.get("bam_gen_s",0), "cli", ["%.3f"%x for x in b["gpu_cli_device_front"]["seconds_all"]], "B %.2f"%b["cpu_B"]["seconds"], "x%.1f"%b.get("speedup_vs_B",0), "pcie", {kk: (round(v,3) if isinstance(v,float) else v) for kk,v in b.get("pcie",{}).items() if kk in ("achieved","frac","frac_of_measured","span_loop_s_all")}, "served", b.get("gpu_cli_served",{}).get("seconds_median"), "identical", b.get("inq_identical"), b.get("size_chosen_by",""))
PY
