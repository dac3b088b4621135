#!/usr/bin/env python3
"""What the launches BEHIND locus_call_small cost when nothing deep is there: the launch sequence of inq_call_batch_device on a
device-resident batch, HIP events around the whole sequence (inq_ctx_timing_read which = 0) and around the first kernel (which = 1),
with the caller's depth hint (one launch) and without it (three: small, mid_walk, the persistent tail).  Round 4 launched ~38
without the hint.

    python3 tools/seq_timing.py [workload] [loci] [reps] [neighbors]

neighbors = k > 0: every locus is offered the reads of its k neighbours on both sides too ((2k + 1) x 30 reads per locus, every read
shared by 2k + 1 loci): k = 1 / 2 / 3 / 4 = 90 / 150 / 210 / 270 reads = the 65 - 256-read path of locus_call_mid_walk (and, at 270,
the walk + tail path).
"""
import json
import os
import sys

ROOT = os.environ.get("INQ_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # INQ_ROOT: another checkout (A/B against an earlier round)
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
    nb = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    dev = torch.device("cuda:0")
    alt = os.environ.get("INQ_LIB")  # another build of libinquistr_hip.so (A/B against an earlier commit)
    ctx = hipcall.Context(0, lib=hipcall.load(alt)) if alt else hipcall.Context(0)
    if os.environ.get("GRID_MEDIUM"):  # A/B of the mid_walk kernel's grid
        ctx.set_option("grid_medium", int(os.environ["GRID_MEDIUM"]))
    shard = synth.DeviceBatch(wl, dev, 0, loci, neighbors=nb)
    stream = torch.cuda.Stream(device=dev)
    depth = wl.reads_per_locus * (2 * nb + 1)
    out = {"workload": name, "loci": loci, "reps": reps, "neighbors": nb, "reads_per_locus": depth, "algorithmic_bytes": shard.algorithmic_bytes()}
    for label, hint in (("hint", depth), ("no_hint", 0), ("hint_again", depth), ("no_hint_again", 0)):
        seq, k = measure(ctx, shard, stream, hint, reps)
        out[label] = {"sequence_ms": seq, "first_kernel_ms": k, "behind_first_kernel_us": (seq - k) * 1e3,
                      "frac_of_8TBps_sequence": shard.algorithmic_bytes() / (seq * 1e-3) / 8e12,
                      "frac_of_8TBps_kernel": shard.algorithmic_bytes() / (k * 1e-3) / 8e12}
    out["no_hint_over_hint"] = min(out["no_hint"]["sequence_ms"], out["no_hint_again"]["sequence_ms"]) / min(out["hint"]["sequence_ms"], out["hint_again"]["sequence_ms"])
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
