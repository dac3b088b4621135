#!/usr/bin/env python3
"""Generates the L2 BAM once (native writer, workload generated on the GPU) and runs the CLI N times with INQ_TIMING=2:
where the end-to-end time goes, run by run.  usage: tools/l2_probe.py [loci] [runs] [extra env K=V ...]"""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools import make_synth_bam

loci = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
extra = dict(kv.split("=", 1) for kv in sys.argv[3:])
tmp = tempfile.mkdtemp(prefix="inq_l2p_")
prefix = os.path.join(tmp, "x")
t = time.time()
make_synth_bam.write_native("unphased100k", loci, prefix, threads=16, device=torch.device("cuda:0"))
print(f"generated in {time.time() - t:.1f} s: {os.path.getsize(prefix + '.bam') / 1e6:.0f} MB", flush=True)
cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "inquistr_amd", "lib", "inquistr")
cmd = [cli, "call", prefix + ".bam", "-R", prefix + ".bed", "-t", "16", "--sample-name", "S", "-u"]
for i in range(runs):
    t = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, env=dict(os.environ, INQ_FRONTEND="device", INQ_TIMING="2", **extra))
    dt = time.perf_counter() - t
    print(f"--- run {i}: {dt:.3f} s rc={r.returncode}")
    print("\n".join(l for l in r.stderr.decode().splitlines() if l.startswith("[inq")), flush=True)
