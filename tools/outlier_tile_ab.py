import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from inquistr_amd import hipcall
ctx = hipcall.Context(0); ctx.timing_enable(True)
for n_cols in (64, 128, 160, 200, 256):
    n_rows = 2_000_000 if n_cols <= 128 else 1_000_000
    rng = np.random.default_rng(1)
    vals = (rng.integers(8, 40, (n_rows, 1)) + rng.integers(-2, 3, (n_rows, n_cols))).astype(np.float32)
    lens = np.full(n_rows, n_cols, dtype=np.uint32)
    out = {}
    for tile in (1, 0):
        ctx.set_option("outlier_tile", tile)
        best = 1e9
        for _ in range(3):
            ctx.timing_reset()
            rc, flags, keep = ctx.outlier_rows(vals, lens, "zscore", minsize=10, zscore_cutoff=3.0, mincluster=5)
            ms, _ = ctx.timing_read(0); best = min(best, ms)
        out[tile] = (best, flags)
    print(f"{n_rows} x {n_cols}: tile {out[1][0]:.3f} ms, transposed copy {out[0][0]:.3f} ms, same flags {np.array_equal(out[1][1], out[0][1])}", flush=True)
