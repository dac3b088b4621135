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
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from inquistr_amd import batch as B
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
        rng = random.Random(seed)
        unphased = bool(i & 1)
        support = rng.choice([1, 2, 3, 3, 5, 40, 700, 9000])
        minlen = rng.choice([5, 0, 12])
        start, end = 700_000, 700_000 + rng.choice([0, 40, 140])
        shapes = gen.random_locus_reads(rng, start, end, rng.choice([12, 60, 200]), long_every=rng.choice([0, 7]))
        # ... and reads that span the window with ONE indel inside it, lengths from a wide range: Calls of many distinct values
        # (the generator's shapes mostly call 0), or of few (ties), by the case
        wide = rng.choice([3, 40, 3000])
        for _ in range(rng.choice([0, 100, 400])):
            pos = start - 10 - rng.randint(1, 300)
            op = rng.choice("IIID")
            ln = rng.randint(1, wide) if op == "I" else rng.randint(1, 30)
            lead = ("S", rng.choice([4, 30])) if rng.random() < 0.1 else None
            cig = ([lead] if lead else []) + [("M", start - pos + rng.randint(0, end - start + 5)), (op, ln), ("M", 400)]
            shapes.append(gen.py.Record(pos=pos if not lead else start + rng.randint(-5, 5), cigar=cig, mapq=60, flag=rng.choice([0, 16])))
        n_pool = rng.choice([21_000, 30_000, 72_000 if i % 4 == 0 else 40_000])
        pool = [(shapes[rng.randrange(len(shapes))], rng.choice([9, 60, 60, 60]), rng.choice([None, 0, 1, 1, 2, 2]), rng.random() < 0.15) for _ in range(n_pool)]
        bb = B.BatchBuilder(minlen=minlen, support=support, unphased=unphased)
        ids = [bb.add_read(r.pos, B.encode_cigar(r.cigar), mapq=mq, phase=ph, reverse=bool(r.flag & 0x10), is_2d=twod) for r, mq, ph, twod in pool]
        order = sorted(range(len(ids)), key=lambda k: (bb._reads[ids[k]][2], k))
        depths = []
        for lo_d, hi_d, cnt in ((1, 64, 6), (65, 256, 5), (257, 2048, 5), (2049, 16384, 3), (16385, min(65536, n_pool), rng.choice([2, 5, 9]))):
            depths += [rng.randint(lo_d, hi_d) for _ in range(cnt)]
        if n_pool > 65_536:
            depths.append(rng.randint(65_537, n_pool))
        depths += [64, 65, 256, 257, 2048, 2049, 16384, 16385][: rng.randint(0, 8)]
        rng.shuffle(depths)
        for j, d in enumerate(depths):
            off = rng.randint(0, n_pool - d)
            sh = rng.choice([-10, 0, 0, 10])  # (windows shifted against each other: not every locus sees the same Calls)
            bb.add_locus(start + sh, end + sh, [ids[k] for k in order[off : off + d]])
        batch = bb.build()
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
