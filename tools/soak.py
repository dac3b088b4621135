#!/usr/bin/env python3
"""Randomised soak of the HIP path against the CPU oracle (GPU box).  Not part of the test-suite: run by hand.
usage: python tools/soak.py [--cases 1500] [--seed0 100000]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inquistr_amd import hipcall
from oracle import orc
from tests import gen

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=1500)
ap.add_argument("--seed0", type=int, default=100000)
a = ap.parse_args()
orc.build()
bad = 0
t0 = time.time()
with hipcall.Context(0) as ctx:
    for i in range(a.cases):
        seed = a.seed0 + i
        unphased = bool(i & 1)
        support = [3, 1, 2, 5, 4][i % 5]
        minlen = [5, 0, 5, 12, 2][(i // 2) % 5]
        max_reads = [40, 40, 70, 130, 300][(i // 7) % 5]
        batch, _ = gen.random_case(seed, n_loci=[25, 60, 8][i % 3], unphased=unphased, minlen=minlen, support=support,
                                   max_reads=max_reads, long_every=[0, 5, 3][i % 3])
        rc, got = ctx.call_batch(batch, debug=True, check=False)
        oc, want = orc.call_batch(batch, debug=True)
        ok = (rc == oc and gen.same_f64(got.phase1, want.phase1) and gen.same_f64(got.phase2, want.phase2)
              and np.array_equal(got.pair_call, want.pair_call) and np.array_equal(got.pair_bits, want.pair_bits)
              and got.n_tie_loci == want.n_tie_loci)
        if not ok:
            bad += 1
            print(f"MISMATCH seed={seed} unphased={unphased} support={support} minlen={minlen} max_reads={max_reads} rc={rc}/{oc}", flush=True)
        if (i + 1) % 250 == 0:
            print(f"{i + 1} cases, {bad} mismatches, {time.time() - t0:.0f}s", flush=True)
print(f"soak done: {a.cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
