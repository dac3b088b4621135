#!/usr/bin/env python3
"""Does bench.py's loop flatter the hot kernel?  It launches over the SAME 2.4 GB batch every step; the chip's memory-side cache
(256 MB) keeps the tail of one step for the head of the next.  Here: the same launch over ONE batch again and again, against the
launch alternating between TWO (or three) different batches of the same shape, so that nothing of a step's data survives to the
next time it is read.  Prints the locus_call_small kernel's HIP-event average for each."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from inquistr_amd import hipcall, synth
from inquistr_amd.batch import InqResultC

wl = synth.WORKLOADS["shard500k"]  # same distribution as unphased100k, 500 000 loci to cut batches from
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
ctx = hipcall.Context(0)
ctx.set_option("max_reads_hint", wl.reads_per_locus)
shards = [synth.DeviceBatch(wl, dev, k * n, (k + 1) * n) for k in range(3)]
out = torch.empty(2, n, dtype=torch.float64, device=dev)
res = InqResultC()
res.phase1, res.phase2 = out[0].data_ptr(), out[1].data_ptr()
res.pair_call = res.pair_bits = None
stream = torch.cuda.Stream(device=dev)
alg = shards[0].algorithmic_bytes()


def run(seq, steps=30, label=""):
    for i in range(6):
        ctx.call_batch_device(shards[seq[i % len(seq)]].c_batch, res, stream.cuda_stream)
    torch.cuda.synchronize()
    ctx.timing_enable(True)
    ctx.timing_reset()
    for i in range(steps):
        ctx.call_batch_device(shards[seq[i % len(seq)]].c_batch, res, stream.cuda_stream)
    torch.cuda.synchronize()
    ms, launches = ctx.timing_read(1)
    ctx.timing_enable(False)
    print(f"{label:46s} {ms / launches * 1e3:8.1f} us per launch = {alg / (ms / launches * 1e-3) / 1e12:5.2f} TB/s of algorithmic bytes ({launches} launches)", flush=True)


print(f"{n} loci per batch, {alg / 1e9:.3f} GB algorithmic per launch")
for rep in range(2):
    run([0], label="the same batch every launch")
    run([0, 1], label="two batches in turn")
    run([0, 1, 2], label="three batches in turn")
ctx.close()
