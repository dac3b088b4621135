#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (written by tools/profile_round.sh on the GPU box) into the small
tracked files under profiles/: <tag>_kernel_stats.csv, <tag>_pmc.csv, <tag>_summary.md and
pmc_latest.json (read by bench.py for roofline.traffic).

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB,
collected in separate --pmc passes; on gfx950 FETCH_SIZE counts exactly half of a 16 B/lane
coalesced stream, so it is doubled; WRITE_SIZE is taken as is.
usage: tools/summarize_profile.py <tag> <workload> <loci_per_gpu> <algorithmic_bytes>"""
import collections
import csv
import json
import os
import statistics
import sys

tag, workload, loci, alg = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def load(f):
    return list(csv.DictReader(open(os.path.join(src, f))))


# kernel_stats: keep our kernels + the header (the rest is torch's generator)
rows = load("kernel_stats.csv")
ours = [r for r in rows if "inq::" in r["Name"]]
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys())
    w.writeheader()
    w.writerows(ours)

pmc = collections.defaultdict(list)
with open(os.path.join(dst, f"{tag}_pmc.csv"), "w", newline="") as f:
    w = None
    for name in ("pmc_fetch.csv", "pmc_write.csv", "pmc_sq.csv"):
        if not os.path.exists(os.path.join(src, name)):
            continue
        for r in load(name):
            if w is None:
                w = csv.DictWriter(f, fieldnames=r.keys())
                w.writeheader()
            w.writerow(r)
            pmc[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))

trace = collections.defaultdict(list)
for r in load("kernel_trace_locus_call.csv"):
    trace[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))

small = [k for k in trace if "locus_call_small" in k][0]
fetch_kib = statistics.mean(pmc[(small, "FETCH_SIZE")])
write_kib = statistics.mean(pmc[(small, "WRITE_SIZE")])
hbm = 2 * fetch_kib * 1024 + write_kib * 1024
avg_ns = statistics.mean(trace[small])
json.dump(
    {
        "workload": workload,
        "loci_per_gpu": loci,
        "kernel": small,
        "hbm_bytes_per_launch": hbm,
        "fetch_size_kib_raw": fetch_kib,
        "write_size_kib_raw": write_kib,
        "avg_kernel_ns_profiled": avg_ns,
        "profile": f"profiles/{tag}_pmc.csv",
        "commit": (os.popen(f"git -C {ROOT} rev-parse --short HEAD 2>/dev/null").read().strip() or None),
        "source": f"profiles/{tag}_pmc.csv: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; "
        "2 x FETCH_SIZE (gfx950 wide-stream correction) + WRITE_SIZE, KiB -> bytes, mean over dispatches",
    },
    open(os.path.join(dst, "pmc_latest.json"), "w"),
    indent=1,
)
with open(os.path.join(dst, f"{tag}_summary.md"), "w") as f:
    f.write(f"# rocprofv3 summary `{tag}` — bench.py --workload {workload} ({loci} loci/GPU)\n\n")
    f.write("Command: `tools/profile_round.sh` = `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-l2` (the driver's command and defaults for the timed region; the legs that start other programs are left out under the profiler)\nplus separate `rocprofv3 --pmc …` passes around `python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-l2`.\n\n")
    f.write("| kernel | dispatches | avg duration (µs) | min | max |\n|---|---|---|---|---|\n")
    for k, v in trace.items():
        f.write(f"| `{k}` | {len(v)} | {statistics.mean(v)/1e3:.1f} | {min(v)/1e3:.1f} | {max(v)/1e3:.1f} |\n")
    bl = os.path.join(src, "trace.log")
    if os.path.exists(bl):
        for line in open(bl):
            if line.startswith("{") and "roofline" in line:
                d = json.loads(line)
                f.write(f"\nbench.py line of the traced run: value {d['value']:.4g} {d['unit']}, roofline.avg_kernel_ms {d['roofline']['avg_kernel_ms']:.4f} "
                        f"(HIP events, {d['roofline']['launches_timed']} timed launches), achieved {d['roofline']['achieved']:.0f} GB/s.\n")
    f.write(f"\nAlgorithmic bytes per launch: {alg} ({alg/1e9:.3f} GB) -> {alg/avg_ns:.0f} GB/s at the profiled duration.\n\n")
    f.write(f"HBM traffic per launch of `{small}`: FETCH_SIZE {fetch_kib:.0f} KiB raw -> x2 (gfx950) = {2*fetch_kib*1024/1e9:.3f} GB read, "
            f"WRITE_SIZE {write_kib:.0f} KiB = {write_kib*1024/1e6:.2f} MB written; total {hbm/1e9:.3f} GB = {hbm/alg:.3f} x algorithmic.\n\n")
    f.write("| counter (mean per dispatch) | " + small + " |\n|---|---|\n")
    for (k, c), v in sorted(pmc.items()):
        if k == small:
            f.write(f"| {c} | {statistics.mean(v):.0f} |\n")
    # the other BASELINE configs, each traced on its own (tools/profile_round.sh): kernel stats next to the bench line of the traced run
    for wl2 in ("phased10k", "expansion50k", "shard500k"):
        ks, bl2 = os.path.join(src, f"kernel_stats_{wl2}.csv"), os.path.join(src, f"bench_line_{wl2}.json")
        if not (os.path.exists(ks) and os.path.exists(bl2)):
            continue
        import shutil

        shutil.copy(ks, os.path.join(dst, f"{tag}_kernel_stats_{wl2}.csv"))
        try:
            d2 = json.loads(open(bl2).read())
        except Exception:  # noqa: BLE001
            continue
        rows2 = [r for r in csv.DictReader(open(ks)) if "locus_call_small" in r["Name"]]
        f.write(f"\n## `--workload {wl2}` ({d2['config']['loci_per_gpu']} loci), `{tag}_kernel_stats_{wl2}.csv`\n\n")
        for r in rows2:
            f.write(f"`{r['Name'].split('(')[0]}`: {r['Calls']} dispatches, avg {float(r['AverageNs'])/1e3:.1f} us\n\n")
        f.write(f"bench.py line of the traced run: value {d2['value']:.4g} loci/s, avg_kernel_ms {d2['roofline']['avg_kernel_ms']:.4f}, "
                f"algorithmic bytes {d2['roofline']['algorithmic_bytes_per_launch']}, frac {d2['roofline']['frac']:.3f}"
                + (f", no-hint sequence {d2['roofline'].get('avg_launch_sequence_ms_no_hint', 0):.4f} ms" if d2['roofline'].get('avg_launch_sequence_ms_no_hint') else "") + ".\n")
print(open(os.path.join(dst, f"{tag}_summary.md")).read())
