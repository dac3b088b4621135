#!/usr/bin/env python3
"""What the launches BEHIND locus_call_small cost when nothing deep is there: the launch sequence of inq_call_batch_device on a
device-resident batch, HIP events around the whole sequence (inq_ctx_timing_read which = 0) and around the first kernel (which = 1),
with the caller's depth hint (one launch) and without it (three: small, mid_walk, the persistent tail).  Round 4 launched ~38
without the hint.

    python3 tools/seq_timing.py [workload] [loci] [reps]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure(ctx, shard, stream, hint, reps, warm=5):
    import torch

    ctx.set_option("max_reads_hint", hint)
    for _ in range(warm):
        ctx.call_batch_device(shard.c_batch, shard.c_result, stream.cuda_stream)
    torch.cuda.synchronize()
    ctx.timing_enable(True)
    ctx.timing_reset()
    for _ in range(reps):
        ctx.call_batch_device(shard.c_batch, shard.c_result, stream.cuda_stream)
    torch.cuda.synchronize()
    seq_ms, n = ctx.timing_read(0)
    k_ms, _ = ctx.timing_read(1)
    ctx.timing_enable(False)
    rc, _ties = ctx.status()
    assert rc == 0, rc
    return seq_ms / n, k_ms / n


def main():
    import torch

    from inquistr_amd import hipcall, synth

    name = sys.argv[1] if len(sys.argv) > 1 else "unphased100k"
    wl = synth.WORKLOADS[name]
    loci = int(sys.argv[2]) if len(sys.argv) > 2 else wl.n_loci
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    dev = torch.device("cuda:0")
    ctx = hipcall.Context(0)
    shard = synth.DeviceBatch(wl, dev, 0, loci)
    stream = torch.cuda.Stream(device=dev)
    out = {"workload": name, "loci": loci, "reps": reps, "algorithmic_bytes": shard.algorithmic_bytes()}
    for label, hint in (("hint", wl.reads_per_locus), ("no_hint", 0), ("hint_again", wl.reads_per_locus), ("no_hint_again", 0)):
        seq, k = measure(ctx, shard, stream, hint, reps)
        out[label] = {"sequence_ms": seq, "first_kernel_ms": k, "behind_first_kernel_us": (seq - k) * 1e3,
                      "frac_of_8TBps_sequence": shard.algorithmic_bytes() / (seq * 1e-3) / 8e12,
                      "frac_of_8TBps_kernel": shard.algorithmic_bytes() / (k * 1e-3) / 8e12}
    out["no_hint_over_hint"] = min(out["no_hint"]["sequence_ms"], out["no_hint_again"]["sequence_ms"]) / min(out["hint"]["sequence_ms"], out["hint_again"]["sequence_ms"])
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
