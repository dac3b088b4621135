#!/usr/bin/env python3
"""L2: end-to-end `call` on a synthetic BAM — the product CLI (C++ sweep front end + HIP kernels)
next to the CPU baseline program oracle/ref_shaped_call in the three reference-shaped modes of
BASELINE.md §3 — with byte-equality of the `.inq` outputs.  Prints one JSON line.
usage: python tools/l2_bench.py [--workload phased10k] [--loci 10000] [--threads 16]"""
import argparse, json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import make_synth_bam

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="phased10k")
ap.add_argument("--loci", type=int, default=10000)
ap.add_argument("--threads", type=int, default=16)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--skip-a", action="store_true")
ap.add_argument("--keep", default="")
a = ap.parse_args()

from inquistr_amd import synth
wl = synth.WORKLOADS[a.workload]
tmp = a.keep or tempfile.mkdtemp(prefix="inq_l2_")
prefix = os.path.join(tmp, f"{a.workload}_{a.loci}")
t0 = time.time()
if not os.path.exists(prefix + ".bam"):
    make_synth_bam.write(a.workload, a.loci, prefix)
gen_s = time.time() - t0
subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref_shaped_call"], stdout=subprocess.DEVNULL)
cli = os.path.join(ROOT, "inquistr_amd", "lib", "inquistr")
ref = os.path.join(ROOT, "oracle", "ref_shaped_call")
un = ["-u"] if wl.unphased else []


def timed(cmd, reps):
    best, out = None, None
    for _ in range(reps):
        t = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True)
        dt = time.perf_counter() - t
        if r.returncode != 0:
            raise SystemExit(f"{cmd} failed: {r.stderr.decode()[-500:]}")
        best = dt if best is None else min(best, dt)
        out = r.stdout
    return best, out

res = {}
t_gpu, out_gpu = timed([cli, "call", prefix + ".bam", "-R", prefix + ".bed", "-t", str(a.threads), "--sample-name", "S"] + un, a.reps + 1)
res["gpu_cli"] = {"seconds": t_gpu, "loci_per_s": a.loci / t_gpu, "threads": a.threads}
for mode, thr in (("C", 1), ("B", a.threads), ("A", a.threads)):
    if mode == "A" and a.skip_a:
        continue
    t, out = timed([ref, prefix + ".bam", prefix + ".bed", mode, str(thr), str(int(wl.unphased)), str(wl.minlen), str(wl.support), "S"], 1 if mode == "A" else a.reps)
    if mode == "C":  # BED order vs sorted: compare as sets of lines
        same = sorted(out.splitlines()) == sorted(out_gpu.splitlines())
    else:
        same = out == out_gpu
    res[f"cpu_{mode}"] = {"seconds": t, "loci_per_s": a.loci / t, "threads": thr, "inq_identical": bool(same)}
print(json.dumps({"level": "L2 end-to-end BAM+BED -> .inq", "workload": a.workload, "loci": a.loci,
                  "bam_mb": os.path.getsize(prefix + ".bam") / 1e6, "bam_gen_s": gen_s, **res,
                  "speedup_vs_A": res["cpu_A"]["seconds"] / t_gpu if "cpu_A" in res else None,
                  "speedup_vs_B": res["cpu_B"]["seconds"] / t_gpu,
                  "note": "CPU modes = oracle/ref_shaped_call: CPU restatement of the reference's control flow, not the Rust binary"}))
