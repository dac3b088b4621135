#!/usr/bin/env python3
"""Small launches (BASELINE config #2 is 10 000 loci): one wave per locus against one WORKGROUP per locus (locus_call_small_split), by
batch size.  HIP events around the first kernel of the sequence, the caller's depth hint set; parity of the two forms checked.

    python3 tools/small_launch.py [workload] [reps]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    from inquistr_amd import hipcall, synth

    name = sys.argv[1] if len(sys.argv) > 1 else "phased10k"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    wl = synth.WORKLOADS[name]
    big = synth.WORKLOADS["unphased100k"] if wl.n_loci < 60_000 else wl
    dev = torch.device("cuda:0")
    ctx = hipcall.Context(0)
    stream = torch.cuda.Stream(device=dev)
    out = {"workload": name, "reps": reps, "sizes": {}}
    for loci in (2_500, 5_000, 10_000, 16_000, 20_000, 24_000, 30_000, 40_000, 60_000, 100_000):
        w = wl if loci <= wl.n_loci else big
        shard = synth.DeviceBatch(w, dev, 0, loci)
        ctx.set_option("max_reads_hint", w.reads_per_locus)
        row = {"workload": w.name, "algorithmic_bytes": shard.algorithmic_bytes()}
        rows = {}
        for label, v in (("one_wave_per_locus", 0), ("one_workgroup_per_locus", 1)):
            ctx.set_option("small_split", v)
            for _ in range(5):
                ctx.call_batch_device(shard.c_batch, shard.c_result, stream.cuda_stream)
            torch.cuda.synchronize()
            ctx.timing_enable(True)
            ctx.timing_reset()
            for _ in range(reps):
                ctx.call_batch_device(shard.c_batch, shard.c_result, stream.cuda_stream)
            torch.cuda.synchronize()
            k_ms, n = ctx.timing_read(1)
            ctx.timing_enable(False)
            assert ctx.status()[0] == 0
            rows[label] = (shard.phase1.clone(), shard.phase2.clone())
            row[label] = {"kernel_us": k_ms / n * 1e3, "frac_of_8TBps": shard.algorithmic_bytes() / (k_ms / n * 1e-3) / 8e12}
        a, b = rows["one_wave_per_locus"], rows["one_workgroup_per_locus"]
        row["rows_identical"] = bool((((a[0] == b[0]) | (a[0].isnan() & b[0].isnan())).all() & ((a[1] == b[1]) | (a[1].isnan() & b[1].isnan())).all()).item())
        out["sizes"][loci] = row
        del shard
    ctx.set_option("small_split", -1)
    print(json.dumps(out))
    for loci, r in out["sizes"].items():
        print(loci, r["workload"], "wave %.1f us %.3f | workgroup %.1f us %.3f | same rows %s" % (
            r["one_wave_per_locus"]["kernel_us"], r["one_wave_per_locus"]["frac_of_8TBps"], r["one_workgroup_per_locus"]["kernel_us"],
            r["one_workgroup_per_locus"]["frac_of_8TBps"], r["rows_identical"]), file=sys.stderr)
    ctx.close()


if __name__ == "__main__":
    main()
