"""Regenerates tests/golden/random_batches.npz: seeded random batches and the CPU ORACLE's outputs
for them (the C restatement in oracle/, cross-checked against oracle/pyoracle.py — NOT outputs of
the Rust reference, which cannot be built here).  Run from the repo root:
    python tests/golden/make_fixtures.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import orc  # noqa: E402
from tests import gen  # noqa: E402


def main():
    out = {}
    cases = [
        dict(seed=101, n_loci=40, unphased=False, minlen=5, support=3, long_every=6),
        dict(seed=102, n_loci=40, unphased=True, minlen=5, support=3, long_every=0),
        dict(seed=103, n_loci=25, unphased=True, minlen=0, support=1, long_every=4, max_reads=90),
        dict(seed=104, n_loci=25, unphased=False, minlen=12, support=5, long_every=0, max_reads=150),
    ]
    for i, kw in enumerate(cases):
        batch, per_locus = gen.random_case(**kw)
        p1, p2, ties = gen.py_expected(batch, per_locus)
        code, res = orc.call_batch(batch, debug=True)
        assert code == 0 and gen.same_f64(res.phase1, p1) and gen.same_f64(res.phase2, p2)
        out[f"c{i}_cigar"] = batch.cigar
        out[f"c{i}_reads"] = batch.reads.view(np.uint8)
        out[f"c{i}_pair_read"] = batch.pair_read
        out[f"c{i}_off"] = batch.locus_pair_off
        out[f"c{i}_start"] = batch.locus_start
        out[f"c{i}_end"] = batch.locus_end
        out[f"c{i}_params"] = np.array([batch.minlen, batch.support, int(batch.unphased), res.n_tie_loci], dtype=np.int64)
        out[f"c{i}_p1"] = res.phase1
        out[f"c{i}_p2"] = res.phase2
        out[f"c{i}_pair_call"] = res.pair_call
        out[f"c{i}_pair_bits"] = res.pair_bits
    out["n_cases"] = np.array(len(cases))
    path = os.path.join(ROOT, "tests", "golden", "random_batches.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
