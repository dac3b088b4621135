#!/usr/bin/env python3
"""L1: rate of the host-buffer entry inq_call_batch (H2D + kernels + D2H over PCIe), pageable and
pinned host buffers.  Reported in DESIGN.md next to the device-resident (L0) number; never bench.py's
`value`.  usage: python tools/l1_bench.py [--workload unphased100k] [--loci 20000] [--reps 5]"""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inquistr_amd import hipcall, synth
from inquistr_amd.batch import Batch, READ_DTYPE

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="unphased100k")
ap.add_argument("--loci", type=int, default=20000)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
wl = synth.WORKLOADS[a.workload]
b = synth.generate_numpy(wl, 0, a.loci)
L = hipcall.load()


def pinned_copy(arr):
    p = C.c_void_p()
    assert L.inq_alloc_pinned(max(arr.nbytes, 1), C.byref(p)) == 0
    buf = (C.c_uint8 * max(arr.nbytes, 1)).from_address(p.value)
    out = np.frombuffer(buf, dtype=np.uint8, count=arr.nbytes).view(arr.dtype)
    out[...] = arr
    return out, p


with hipcall.Context(0) as ctx:
    def run(batch, label):
        ctx.call_batch(batch)  # warm-up (allocations)
        best = 1e9
        for _ in range(a.reps):
            t0 = time.perf_counter(); ctx.call_batch(batch); best = min(best, time.perf_counter() - t0)
        nbytes = batch.cigar.nbytes + batch.reads.nbytes + batch.pair_read.nbytes
        print(f"{label}: {batch.n_loci / best:.3e} loci/s  ({best*1e3:.2f} ms for {batch.n_loci} loci, {nbytes/best/1e9:.1f} GB/s host->device incl. kernels)")

    run(b, "pageable host buffers")
    keep = []
    arrs = {}
    for name in ("cigar", "reads", "pair_read", "locus_pair_off", "locus_start", "locus_end"):
        arrs[name], p = pinned_copy(getattr(b, name)); keep.append(p)
    pb = Batch(minlen=b.minlen, support=b.support, unphased=b.unphased, **arrs)
    run(pb, "pinned host buffers  ")
    for p in keep:
        L.inq_free_pinned(p)
