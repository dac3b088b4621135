#!/usr/bin/env python3
"""A/B of one host-library environment switch on a RESIDENT context: a session calls the same file N times with the switch at each
of its values in turn (GPU box; by hand).  usage: python tools/session_ab.py NAME=v0,v1 workload loci seq(0/1) [runs]   (a value "-" = the variable unset)"""
import os, statistics, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
name, vals = sys.argv[1].split("=")
vals = vals.split(",")
wl, loci, seq = sys.argv[2], int(sys.argv[3]), sys.argv[4] == "1"
runs = int(sys.argv[5]) if len(sys.argv) > 5 else 8
import torch
from inquistr_amd import call, synth
from tools import make_synth_bam
d = tempfile.mkdtemp(prefix="inq_ab_")
prefix = os.path.join(d, "f")
make_synth_bam.write_native(wl, loci, prefix, threads=16, device=torch.device("cuda", 0), seq=seq, level=6)
for _ in range(2):
    with open(prefix + ".bam", "rb", buffering=0) as f:
        while f.read(64 << 20):
            pass
print(f"{wl} {loci} loci seq={seq}: {os.path.getsize(prefix + '.bam') / 1e6:.0f} MB")
un = synth.WORKLOADS[wl].unphased
res = {v: [] for v in vals}


def setv(v):  # "-" = the variable unset
    if v == "-":
        os.environ.pop(name, None)
    else:
        os.environ[name] = v


with call.Session(0) as S, open(os.devnull, "w") as out:
    for v in vals:  # warm
        setv(v)
        S.call(prefix + ".bam", region_file=prefix + ".bed", threads=16, unphased=un, sample_name="S", out=out, frontend="device")
    for r in range(runs):
        for v in vals:
            setv(v)
            t = time.perf_counter()
            S.call(prefix + ".bam", region_file=prefix + ".bed", threads=16, unphased=un, sample_name="S", out=out, frontend="device")
            res[v].append(time.perf_counter() - t)
for v in vals:
    x = sorted(res[v])
    print(f"  {name}={v}: median {statistics.median(x) * 1e3:.1f} ms  min {x[0] * 1e3:.1f}  max {x[-1] * 1e3:.1f}  ({runs} calls)")
