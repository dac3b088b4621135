#!/usr/bin/env python3
"""A/B of ctx options in ONE process, interleaved rounds (cdna_hip_programming.md §5.4 rule 24).
usage: python tools/ab_options.py key=v1,v2 [--workload unphased100k] [--rounds 8] [--steps 10]"""
import argparse, os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from inquistr_amd import hipcall, synth

ap = argparse.ArgumentParser()
ap.add_argument("spec")
ap.add_argument("--workload", default="unphased100k")
ap.add_argument("--loci", type=int, default=0)
ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--neighbors", type=int, default=0)
ap.add_argument("--libs", default="", help="comma-separated library files: A/B of two BUILDS (the option spec is then applied to each)")
ap.add_argument("--which", type=int, default=1, help="0 = whole launch sequence, 1 = locus_call_small only")
ap.add_argument("--reads-per-locus", type=int, default=0)
a = ap.parse_args()
key, vals = a.spec.split("=")
vals = [int(v) for v in vals.split(",")]
wl = synth.WORKLOADS[a.workload]
if a.reads_per_locus:
    import dataclasses
    wl = dataclasses.replace(wl, reads_per_locus=a.reads_per_locus)
dev = torch.device("cuda:0")
d = synth.DeviceBatch(wl, dev, 0, a.loci or wl.n_loci, neighbors=a.neighbors)
print(f"pairs {d.n_pairs} reads {d.n_reads} algorithmic GB {d.algorithmic_bytes()/1e9:.3f}")
libs = [l for l in a.libs.split(",") if l] or [None]
ctxs = [(os.path.basename(l) if l else "default", hipcall.Context(0, lib=hipcall.load(l) if l else None)) for l in libs]
st = torch.cuda.current_stream().cuda_stream
res = {(n, v): [] for n, _ in ctxs for v in vals}
for _, c in ctxs:
    c.timing_enable(True)
for r in range(a.rounds + 1):
    for name, ctx in ctxs:
        for v in vals:
            ctx.set_option(key, v)
            ctx.timing_reset()
            for _ in range(a.steps):
                ctx.call_batch_device(d.c_batch, d.c_result, st)
            torch.cuda.synchronize()
            ms, n = ctx.timing_read(a.which)
            if r:  # round 0 = warm-up
                res[(name, v)].append(ms / n)
for _, c in ctxs:
    assert c.status()[0] == 0
ab = d.algorithmic_bytes()
for (name, v), t in res.items():
    m = statistics.median(t); mn = min(t)
    print(f"{name} {key}={v}: median {m*1e3:.1f} us  min {mn*1e3:.1f} us  -> {ab/m/1e6:.0f} GB/s median, {ab/mn/1e6:.0f} GB/s best")
