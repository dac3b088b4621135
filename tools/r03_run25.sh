#!/bin/bash
# GPU box, round 3: the default bench line with the measured read ceiling and the 12.8 GB SEQ-bearing block; wall time of the whole command.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03b9
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
t0=$(date +%s.%N)
timeout -k 10 600 python3 -m pytest tests/test_gpu_end_to_end.py -m gpu -x -q > $OUT/gputest_e2e.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/gputest_e2e.log
timeout -k 10 900 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $?"
t1=$(date +%s.%N)
python3 -c "print(f\"bench.py default: {$t1 - $t0:.1f} s\")" | tee $OUT/bench_wall.txt
python3 - <<'PY'
import json, os
d = json.loads(open(os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r03b9/bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]; print({k: r.get(k) for k in ("achieved", "frac", "frac_of_measured")}, r.get("peak_measured", {}).get("all"))
for k in ("l2", "l2_seq", "l2_seq_large"):
    b = d.get(k, {})
    print(k, {x: b.get(x) for x in ("bam_mb", "bam_gen_s", "speedup_vs_B", "speedup_vs_B_all_cores", "inq_identical", "error", "skipped")}, b.get("gpu_cli_device_front", {}).get("seconds_all"), b.get("cpu_B", {}).get("seconds"))
print(d.get("l2_seq_large", {}).get("device_front_stages"))
PY
