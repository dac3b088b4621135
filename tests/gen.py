"""Seeded random test-case generator: python Records per locus + the packed Batch.

Shapes follow the edge cases the reference's semantics make interesting:
reads ending exactly on window edges, fully-inside reads, low mapq, missing / zero HP,
soft clips with and without a 2D supplementary alignment, every CIGAR op, empty CIGARs,
unmapped-but-placed reads, reads shared by neighbouring loci, empty loci, long CIGARs.
"""
from __future__ import annotations

import random
from typing import List, Tuple

import numpy as np

from inquistr_amd.batch import Batch, BatchBuilder, encode_cigar
from oracle import pyoracle as py


def random_cigar(rng: random.Random, n_ops: int, big: bool = False) -> List[Tuple[str, int]]:
    ops = []
    for i in range(n_ops):
        r = rng.random()
        if i % 2 == 0 or r < 0.15:
            op = rng.choice("MMMMMM=X")
            ln = rng.randint(1, 60)
        elif r < 0.50:
            op, ln = "I", rng.choice([1, 2, 3, 5, 6, 7, 12, 40, 300])
        elif r < 0.85:
            op, ln = "D", rng.choice([1, 2, 4, 5, 6, 9, 15, 33])
        elif r < 0.90:
            op, ln = "N", rng.randint(1, 80)
        elif r < 0.94:
            op, ln = "S", rng.choice([3, 5, 6, 20, 150])
        elif r < 0.97:
            op, ln = "H", rng.randint(1, 50)
        else:
            op, ln = "P", rng.randint(1, 5)
        if big and rng.random() < 0.02:
            op, ln = "I", rng.randint(5000, 50000)
        ops.append((op, ln))
    return ops


def random_locus_reads(rng: random.Random, start: int, end: int, n: int, long_every: int = 0) -> List[py.Record]:
    start_ext, end_ext = start - 10, end + 10
    recs = []
    for k in range(n):
        style = rng.random()
        if style < 0.55:  # spanning read
            pos = start_ext - rng.randint(0, 400)
        elif style < 0.65:  # starts exactly on / next to the window edge
            pos = start_ext + rng.choice([-1, 0, 1])
        elif style < 0.80:  # starts inside the window
            pos = rng.randint(start_ext, end_ext)
        elif style < 0.90:  # far left, may or may not reach
            pos = start_ext - rng.randint(300, 900)
        else:  # at / beyond the right edge
            pos = end_ext + rng.choice([-2, -1, 0, 1, 50])
        pos = max(pos, 0)
        n_ops = rng.choice([0, 1, 2, 3, 5, 9, 17, 30, 63, 64, 65])
        if long_every and k % long_every == 0:
            n_ops = rng.choice([255, 256, 257, 300, 511, 513, 700, 1500])
        cig = random_cigar(rng, n_ops, big=rng.random() < 0.1)
        # sometimes force the read to end exactly on the right edge of the window
        if cig and rng.random() < 0.15:
            rlen = sum(l for o, l in cig if o in "MDN=X")
            want = end_ext - pos + rng.choice([-1, 0, 0, 1])
            if want > rlen:
                cig.append(("M", want - rlen))
        flag = 0
        if rng.random() < 0.3:
            flag |= 0x10
        if rng.random() < 0.03:
            flag |= 0x4
        hp = None
        r = rng.random()
        if r < 0.40:
            hp = ("C", 1)
        elif r < 0.80:
            hp = ("C", 2)
        elif r < 0.88:
            hp = ("C", 0)
        elif r < 0.92:
            hp = ("i", rng.choice([1, 2, 257, 258]))
        sa = None
        if rng.random() < 0.25:
            strand = rng.choice("+-")
            sa_pos = pos + rng.randint(-200, 200)
            entry = f"chr7,{sa_pos},{strand},{rng.randint(1, 300)}M{rng.randint(1, 50)}S,60,0;"
            if rng.random() < 0.15:
                entry += "chr2,100,+,50M,0,0;"
            sa = ("Z", entry)
        mapq = rng.choice([0, 5, 10, 11, 20, 60, 60, 60, 60])
        recs.append(py.Record(pos=pos, cigar=cig, mapq=mapq, flag=flag, hp=hp, sa=sa))
    return recs


def random_case(seed: int, n_loci: int = 40, unphased: bool = False, minlen: int = 5, support: int = 3,
                max_reads: int = 40, long_every: int = 0, share: bool = True):
    """Returns (Batch, per-locus python records)."""
    rng = random.Random(seed)
    bb = BatchBuilder(minlen=minlen, support=support, unphased=unphased)
    per_locus: List[List[py.Record]] = []
    cursor = 2000
    prev: List[Tuple[int, py.Record]] = []
    for j in range(n_loci):
        start = cursor + rng.randint(0, 300)
        end = start + rng.randint(0, 250)
        cursor = end + rng.randint(30, 1500)
        n = rng.choice([0, 1, 2, 3, 5, 6, 7, 12, 20, max_reads])
        if j == 1 and max_reads > 40:
            n = max_reads  # deep-locus tests need the depth they ask for
        recs = random_locus_reads(rng, start, end, n, long_every)
        idx = []
        merged: List[py.Record] = []
        # neighbouring loci share some reads (a read overlapping k loci is offered k times)
        if share and prev and rng.random() < 0.5:
            for ri, r in prev[: rng.randint(1, 4)]:
                idx.append(ri)
                merged.append(r)
        cur = []
        for r in recs:
            phase = py.get_phase(r)
            ri = bb.add_read(
                pos=r.pos,
                cigar_words=encode_cigar(r.cigar),
                mapq=r.mapq,
                phase=phase,
                reverse=bool(r.flag & 0x10),
                unmapped=bool(r.flag & 0x4),
                is_2d=py.is_accidental_2d(r),
            )
            idx.append(ri)
            merged.append(r)
            cur.append((ri, r))
        # "file order" = increasing position, ties by insertion (like a coordinate-sorted BAM)
        order = sorted(range(len(idx)), key=lambda k: (merged[k].pos, idx[k]))
        idx = [idx[k] for k in order]
        merged = [merged[k] for k in order]
        bb.add_locus(start, end, idx)
        per_locus.append(merged)
        prev = cur
    return bb.build(), per_locus


def py_expected(batch: Batch, per_locus) -> Tuple[np.ndarray, np.ndarray, int]:
    p1 = np.full(batch.n_loci, np.nan)
    p2 = np.full(batch.n_loci, np.nan)
    ties = 0
    for j, recs in enumerate(per_locus):
        s, e = int(batch.locus_start[j]), int(batch.locus_end[j])
        if batch.unphased:
            a, b, t = py.genotype_repeat_unphased(recs, 0, s, e, batch.minlen, batch.support)
            ties += int(t)
        else:
            a, b = py.genotype_repeat_phased(recs, 0, s, e, batch.minlen, batch.support)
        p1[j], p2[j] = a, b
    return p1, p2, ties


def same_f64(a: np.ndarray, b: np.ndarray) -> bool:
    """Bit-exact up to NaN payload: NaNs must coincide, everything else must be equal."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and bool(np.all((np.isnan(a) & np.isnan(b)) | (a == b)))


def mixed_depth_case(seed: int, case_index: int = 0):
    """One batch that mixes the depth classes of DESIGN.md 3.2 - <= 64 offered reads, 65 - 256, 257 - 2 048 (reduced by the workgroup
    that walked them), 2 049 - 16 384, several loci of 16 385 - 65 536 (walked by a group of workgroups each) and, every fourth
    case_index, one locus beyond 65 536 (the whole grid) -, reads drawn from a pool of shapes (ties everywhere) plus reads with ONE
    indel of a wide range of lengths inside the window (Calls of many distinct values), HP / mapq / strand / 2D bits random per read,
    `support` from 1 to beyond a group's size.  Returns (Batch, depths)."""
    rng = random.Random(seed)
    unphased = bool(case_index & 1)
    support = rng.choice([1, 2, 3, 3, 5, 40, 700, 9000])
    minlen = rng.choice([5, 0, 12])
    start, end = 700_000, 700_000 + rng.choice([0, 40, 140])
    shapes = random_locus_reads(rng, start, end, rng.choice([12, 60, 200]), long_every=rng.choice([0, 7]))
    wide = rng.choice([3, 40, 3000])
    for _ in range(rng.choice([0, 100, 400])):
        pos = start - 10 - rng.randint(1, 300)
        op = rng.choice("IIID")
        ln = rng.randint(1, wide) if op == "I" else rng.randint(1, 30)
        lead = ("S", rng.choice([4, 30])) if rng.random() < 0.1 else None
        cig = ([lead] if lead else []) + [("M", start - pos + rng.randint(0, end - start + 5)), (op, ln), ("M", 400)]
        shapes.append(py.Record(pos=pos if not lead else start + rng.randint(-5, 5), cigar=cig, mapq=60, flag=rng.choice([0, 16])))
    n_pool = 72_000 if case_index % 4 == 0 else rng.choice([21_000, 30_000, 40_000])
    pool = [(shapes[rng.randrange(len(shapes))], rng.choice([9, 60, 60, 60]), rng.choice([None, 0, 1, 1, 2, 2]), rng.random() < 0.15) for _ in range(n_pool)]
    bb = BatchBuilder(minlen=minlen, support=support, unphased=unphased)
    ids = [bb.add_read(r.pos, encode_cigar(r.cigar), mapq=mq, phase=ph, reverse=bool(r.flag & 0x10), is_2d=twod) for r, mq, ph, twod in pool]
    order = sorted(range(len(ids)), key=lambda k: (bb._reads[ids[k]][2], k))
    depths = []
    for lo_d, hi_d, cnt in ((1, 64, 6), (65, 256, 5), (257, 2048, 5), (2049, 16384, 3), (16385, min(65536, n_pool), rng.choice([2, 5, 9]))):
        depths += [rng.randint(lo_d, hi_d) for _ in range(cnt)]
    if n_pool > 65_536:
        depths.append(rng.randint(65_537, n_pool))
    depths += [64, 65, 256, 257, 2048, 2049, 16384, 16385][: rng.randint(0, 8)]
    rng.shuffle(depths)
    for d in depths:
        off = rng.randint(0, n_pool - d)
        sh = rng.choice([-10, 0, 0, 10])  # (windows shifted against each other: not every locus sees the same Calls)
        bb.add_locus(start + sh, end + sh, [ids[k] for k in order[off : off + d]])
    return bb.build(), depths
