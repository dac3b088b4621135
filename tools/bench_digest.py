#!/usr/bin/env python3
"""A few lines out of one bench.py JSON line (the default line is ~20 KB)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.4g %s  roofline frac %.3f  of measured %s" % (d["value"], d["unit"], d["roofline"]["frac"], d["roofline"].get("frac_of_measured")))
print("startup_floor", d.get("startup_floor", {}).get("seconds_all"), " h2d GB/s", d.get("h2d_copy_peak", {}).get("value"))
print("l1", {k: d.get("l1", {}).get(k) for k in ("loci_per_s", "GBps_host_to_device_incl_kernels", "error")})
r = d["roofline"]
print("no hint: sequence %s ms, frac %s, over hinted %s" % (r.get("avg_launch_sequence_ms_no_hint"), r.get("frac_no_hint"), r.get("no_hint", {}).get("over_hinted_sequence")))
for name, c in d.get("l0_configs", {}).items():
    print("l0", name, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in c.items() if k in ("loci", "value", "avg_kernel_ms", "frac", "error", "what")})
for k in ("l2", "l2_seq", "l2_phased", "l2_seq_large"):
    b = d.get(k, {})
    if not b or "error" in b or "skipped" in b:
        print(k, b)
        continue
    g = b["gpu_cli_device_front"]
    pc = b.get("pcie", {})
    print(k, "loci", b.get("loci"), "bam_mb %.0f" % b.get("bam_mb", 0), "level", b.get("zlib_level"), "gen_s %.1f" % b.get("bam_gen_s", 0),
          "| cli", ["%.3f" % x for x in g["seconds_all"]], "b2b", ["%.3f" % x for x in g.get("seconds_back_to_back", [])],
          "| B %.2f s x%.1f" % (b["cpu_B"]["seconds"], b.get("speedup_vs_B", 0)),
          "| pcie %.1f GB/s frac_of_measured %s loops %s" % (pc.get("achieved", 0), pc.get("frac_of_measured") and round(pc["frac_of_measured"], 3), [round(x, 4) for x in pc.get("span_loop_s_all", [])]),
          "| served", b.get("gpu_cli_served", {}).get("seconds_median"), "| identical", b.get("inq_identical"), b.get("size_chosen_by", ""))
    if "written" in b:
        print("   written:", {k: (round(v, 1) if isinstance(v, float) else v) for k, v in b["written"].items() if k != "how"})
