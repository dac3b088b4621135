"""Kernel time of inq_outlier_rows (HIP events, inq_ctx_timing_read) on a synthetic cohort matrix.
usage: python tools/outlier_bench.py [n_rows] [n_cols] [zscore|dbscan]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inquistr_amd import hipcall  # noqa: E402


def main():
    n_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    n_cols = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    method = sys.argv[3] if len(sys.argv) > 3 else "zscore"
    rng = np.random.default_rng(1)
    vals = (rng.integers(8, 40, (n_rows, 1)) + rng.integers(-2, 3, (n_rows, n_cols))).astype(np.float32)
    hit = rng.integers(0, n_cols, n_rows)
    vals[np.arange(n_rows), hit] *= rng.choice([1.0, 1.0, 6.0], n_rows).astype(np.float32)  # a third of the loci carry an expansion
    vals[rng.random((n_rows, n_cols)) < 0.03] = np.nan
    lens = np.full(n_rows, n_cols, dtype=np.uint32)
    ctx = hipcall.Context(0)
    ctx.timing_enable(True)
    for _ in range(3):
        ctx.timing_reset()
        t = time.perf_counter()
        rc, flags, keep = ctx.outlier_rows(vals, lens, method, minsize=10, zscore_cutoff=3.0, mincluster=max(1, n_cols.bit_length() - 1))
        wall = time.perf_counter() - t
        ms, _ = ctx.timing_read(0)
        cells = n_rows * n_cols
        print(f"{method}: {n_rows} loci x {n_cols} values: kernel {ms:.3f} ms = {n_rows / ms / 1e3:.1f} M loci/s, "
              f"{cells * 5 / ms / 1e6:.1f} GB/s of matrix read + flags written once (call incl. PCIe {wall * 1e3:.0f} ms); "
              f"{int(keep.sum())} loci kept, {int(flags.sum())} outlying values", flush=True)
    # CPU baseline: the C restatement (oracle/outlier_oracle.c, OpenMP over loci) on a sample of the same matrix, and
    # a check that it reports the same values
    from oracle import outlier_oracle as oo

    threads = min(16, len(os.sched_getaffinity(0)))
    sample = min(n_rows, max(1000, (200_000_000 if method == "zscore" else 4_000_000_000) // max(n_cols * (1 if method == "zscore" else n_cols), 1)))
    best = None
    for _ in range(3):
        t = time.perf_counter()
        cf, ck = oo.c_outlier_rows(vals[:sample], lens[:sample], method, 10, 3.0, max(1, n_cols.bit_length() - 1), threads)
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    same = bool((cf == flags[:sample]).all() and (ck == keep[:sample]).all())
    print(f"cpu baseline ({threads} threads, first {sample} loci, C restatement of src/outlier.rs, not the Rust binary): "
          f"{best * 1e3:.1f} ms = {sample / best / 1e6:.2f} M loci/s; same flags as the GPU: {same}", flush=True)


if __name__ == "__main__":
    main()
