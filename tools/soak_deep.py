#!/usr/bin/env python3
"""Randomised soak of the DEEP-locus kernels (csrc/kernels.hip locus_call_mid_walk, csrc/deep_select.hip locus_call_tail) against the
CPU oracle (GPU box; not part of the test-suite: run by hand).  Every case is one batch that mixes the depth classes of DESIGN.md 3.2
- <= 64 offered reads, 65 - 256, 257 - 2 048 (reduced by the workgroup that walked them), 2 049 - 16 384, 16 385 - 65 536 (several
of them: walked by a group of workgroups each) and, every few cases, one locus beyond 65 536 (the whole grid) -, reads drawn from a
pool of shapes so that equal Calls are everywhere, HP / mapq / strand / 2D bits random per read, `support` from 1 to beyond a group's
size, both modes, with and without the caller's depth hint.  Rows, per-pair Calls and bits, tie counts must equal the oracle's.
usage: python tools/soak_deep.py [--cases 60] [--seed0 500000]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from inquistr_amd import hipcall
from oracle import orc
from tests import gen

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=60)
ap.add_argument("--seed0", type=int, default=500000)
a = ap.parse_args()
orc.build()
bad = 0
t0 = time.time()
with hipcall.Context(0) as ctx:
    for i in range(a.cases):
        seed = a.seed0 + i
        batch, depths = gen.mixed_depth_case(seed, i)
        unphased, support, minlen = bool(batch.unphased), int(batch.support), int(batch.minlen)
        hint = 0 if i % 3 else int(max(depths))
        ctx.set_option("max_reads_hint", hint)
        rc, got = ctx.call_batch(batch, debug=True, check=False)
        oc, want = orc.call_batch(batch, debug=True, threads=8)
        ok = (rc == oc and gen.same_f64(got.phase1, want.phase1) and gen.same_f64(got.phase2, want.phase2)
              and np.array_equal(got.pair_call, want.pair_call) and np.array_equal(got.pair_bits, want.pair_bits)
              and got.n_tie_loci == want.n_tie_loci)
        if not ok:
            bad += 1
            where = np.nonzero(~((np.isnan(got.phase1) & np.isnan(want.phase1)) | (got.phase1 == want.phase1)) |
                               ~((np.isnan(got.phase2) & np.isnan(want.phase2)) | (got.phase2 == want.phase2)))[0]
            print(f"MISMATCH seed={seed} unphased={unphased} support={support} minlen={minlen} hint={hint} rc={rc}/{oc} loci {where[:8].tolist()} "
                  f"depths {[depths[k] for k in where[:8]]}", flush=True)
        if (i + 1) % 10 == 0:
            print(f"{i + 1} cases, {bad} mismatches, {time.time() - t0:.0f}s ({batch.n_pairs} pairs, {int(np.sum(~np.isnan(got.phase1)))} of {batch.n_loci} rows numeric)", flush=True)
    ctx.set_option("max_reads_hint", 0)
print(f"deep soak done: {a.cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
