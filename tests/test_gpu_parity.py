"""Parity of the HIP path (through the C ABI) against the CPU oracle.  Needs an MI355X.

Bit-exact bar: integer calls and flags equal, medians equal as f64 (NaN where the oracle has NaN).
"""
import json
import os

import numpy as np
import pytest

from inquistr_amd import batch as B
from inquistr_amd import synth
from tests import gen

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from inquistr_amd import hipcall

    c = hipcall.Context(0)
    assert c.backend.startswith("hip:gfx950")
    yield c
    c.close()


def _assert_same(got, want, what=""):
    assert gen.same_f64(got.phase1, want.phase1), f"phase1 differs {what}"
    assert gen.same_f64(got.phase2, want.phase2), f"phase2 differs {what}"
    if want.pair_call is not None and got.pair_call is not None:
        bad = np.nonzero(got.pair_call != want.pair_call)[0]
        assert bad.size == 0, f"pair_call differs at {bad[:8]} {what}"
        bad = np.nonzero(got.pair_bits != want.pair_bits)[0]
        assert bad.size == 0, f"pair_bits differs at {bad[:8]}: {got.pair_bits[bad[:8]]} vs {want.pair_bits[bad[:8]]} {what}"
    assert got.n_tie_loci == want.n_tie_loci, what


def test_kat_loci(ctx, orc, kat):
    """The hand-derived locus vectors of tests/golden/kat_call.json, one batch per mode."""
    from oracle import pyoracle as py
    from tests.test_oracle_kat import _num, _pyrec

    for mode in ("phased", "unphased"):
        cases = [v for v in kat["loci"] if v["mode"] == mode]
        by_params = {}
        for v in cases:
            by_params.setdefault((v["minlen"], v["support"]), []).append(v)
        for (minlen, support), vs in by_params.items():
            bb = B.BatchBuilder(minlen=minlen, support=support, unphased=(mode == "unphased"))
            for v in vs:
                idx = []
                for r in v["reads"]:
                    rec = _pyrec(r)
                    idx.append(
                        bb.add_read(rec.pos, B.encode_cigar(rec.cigar), mapq=rec.mapq, phase=py.get_phase(rec),
                                    reverse=bool(rec.flag & 0x10), unmapped=bool(rec.flag & 0x4),
                                    is_2d=py.is_accidental_2d(rec))
                    )
                bb.add_locus(v["start"], v["end"], idx)
            batch = bb.build()
            rc, got = ctx.call_batch(batch, debug=True)
            assert rc == 0
            for j, v in enumerate(vs):
                want = [_num(x) for x in v["expect"]]
                assert gen.same_f64(np.array([got.phase1[j], got.phase2[j]]), np.array(want)), v["name"]
            assert got.n_tie_loci == sum(1 for v in vs if v.get("tie"))
            _assert_same(got, orc.call_batch(batch, debug=True)[1], mode)


def test_kat_call_from_cigar(ctx, kat):
    """Every call_from_cigar vector as a one-read locus; the per-pair Call comes back through
    the debug outputs of the ABI."""
    from oracle import pyoracle as py
    from tests.test_oracle_kat import _pyrec

    groups = {}
    for v in kat["call_from_cigar"]:
        groups.setdefault(v["minlen"], []).append(v)
    for minlen, vs in groups.items():
        bb = B.BatchBuilder(minlen=minlen, support=1, unphased=False)
        for v in vs:
            rec = _pyrec(v)
            ri = bb.add_read(rec.pos, B.encode_cigar(rec.cigar), phase=1, reverse=bool(rec.flag & 0x10),
                             is_2d=py.is_accidental_2d(rec))
            bb.add_locus(v["start"] + 10, v["end"] - 10, [ri])
        batch = bb.build()
        rc, got = ctx.call_batch(batch, debug=True)
        assert rc == 0
        for j, v in enumerate(vs):
            kind, val = v["expect"]
            assert got.pair_call[j] == val, v["name"]
            assert bool(got.pair_bits[j] & B.INQ_PAIR_CLIP) == (kind == "Clip"), v["name"]


@pytest.mark.parametrize("seed", range(48))
@pytest.mark.parametrize("unphased", [False, True])
def test_random_vs_oracle(ctx, orc, seed, unphased):
    support = [3, 1, 2, 5][seed % 4]
    minlen = [5, 0, 5, 12][seed % 4]
    batch, _ = gen.random_case(seed, n_loci=60, unphased=unphased, minlen=minlen, support=support,
                               long_every=5 if seed % 2 == 0 else 0)
    rc, got = ctx.call_batch(batch, debug=True)
    oc, want = orc.call_batch(batch, debug=True)
    assert rc == oc == 0
    _assert_same(got, want, f"seed={seed} unphased={unphased}")


@pytest.mark.parametrize("unphased", [False, True])
@pytest.mark.parametrize("max_reads", [64, 65, 130, 256, 257, 700, 2300, 8192, 8300, 16384, 16500])
def test_deep_loci(ctx, orc, unphased, max_reads):
    """Loci with more than 64 offered reads take the work-list kernel."""
    # <= 64: wave per locus; <= 256: four reads per lane; <= 16384: walk kernel + LDS sort (3 size classes); beyond: global ranks
    batch, _ = gen.random_case(1000 + max_reads, n_loci=24 if max_reads < 2000 else 6, unphased=unphased,
                               max_reads=max_reads, long_every=9, support=3)
    assert int(np.diff(batch.locus_pair_off.astype(np.int64)).max()) >= max_reads
    rc, got = ctx.call_batch(batch, debug=True)
    oc, want = orc.call_batch(batch, debug=True)
    assert rc == oc == 0
    _assert_same(got, want, f"max_reads={max_reads}")


def test_clip_heavy_ties(ctx, orc):
    """Many equal values of mixed Span/Clip: exercises the clip ranking and the tie counter."""
    import random

    rng = random.Random(7)
    for unphased in (False, True):
        bb = B.BatchBuilder(minlen=5, support=3, unphased=unphased)
        for j in range(200):
            start = 5000 + 1000 * j
            end = start + 40
            idx = []
            for k in range(rng.choice([4, 7, 12, 33, 64, 90])):
                v = rng.choice([8, 8, 8, 20, 20, 31])
                if rng.random() < 0.5:
                    cig = [("M", 20), ("S", v), ("M", 200)]  # Clip(v) at refpos start_ext+21... inside window
                    pos = start - 10
                else:
                    cig = [("M", 100), ("I", v), ("M", 200)]
                    pos = start - 10 - 80
                idx.append(bb.add_read(pos, B.encode_cigar(cig), phase=rng.choice([1, 2])))
            bb.add_locus(start, end, idx)
        batch = bb.build()
        rc, got = ctx.call_batch(batch, debug=True)
        oc, want = orc.call_batch(batch, debug=True)
        assert rc == oc == 0
        if unphased:
            assert want.n_tie_loci > 0
        _assert_same(got, want, f"unphased={unphased}")


@pytest.mark.parametrize("unphased", [False, True])
def test_huge_values(ctx, orc, unphased):
    """Calls beyond 25 and beyond 32 bits: the i64 ranking path (28-bit op lengths, several per read)."""
    import random

    rng = random.Random(11)
    big = (1 << 28) - 1
    bb = B.BatchBuilder(minlen=5, support=2, unphased=unphased)
    for j in range(40):
        start = 10_000 + 3000 * j
        idx = []
        for k in range(rng.choice([4, 9, 16, 31])):
            n_big = rng.choice([0, 1, 1, 3, 9, 17, 40])
            cig = [("M", 100)]
            for i in range(n_big):
                # insertions do not advance the reference, so all of them start inside the window; a
                # deletion may only come last (it moves everything behind it out of the window)
                op = "D" if (i == n_big - 1 and rng.random() < 0.4) else "I"
                cig += [(op, big if n_big == 40 else rng.choice([big, big - 1, 1 << 24, (1 << 24) - 1, 1 << 26])), ("M", 1)]
            cig += [("I", rng.choice([6, 7, 8])), ("M", 300)]
            idx.append(bb.add_read(start - 10 - 80, B.encode_cigar(cig), phase=1 + (k & 1),
                                   is_2d=False))
        bb.add_locus(start, start + 60, idx)
    batch = bb.build()
    rc, got = ctx.call_batch(batch, debug=True)
    oc, want = orc.call_batch(batch, debug=True)
    assert rc == oc == 0
    assert np.abs(want.pair_call).max() > (1 << 32)
    _assert_same(got, want, f"huge unphased={unphased}")


@pytest.mark.parametrize("unphased", [False, True])
def test_wide_windows_fill_the_lane_queue(ctx, orc, unphased):
    """Loci tens of kb wide with reads made of thousands of short ops: nearly every lane of every chunk
    starts inside the window, so the LDS window-lane queue drains many times per read (and in the middle
    of multi-chunk reads), with soft clips and accidental-2D reads mixed in."""
    import random

    rng = random.Random(21)
    bb = B.BatchBuilder(minlen=2, support=2, unphased=unphased)
    for j in range(12):
        start = 100_000 + 200_000 * j
        end = start + rng.choice([300, 5_000, 40_000])
        idx = []
        for k in range(rng.choice([3, 8, 20, 70])):
            ops = []
            for _ in range(rng.choice([40, 300, 1200, 3000])):
                ops.append(("M", rng.randint(1, 30)))
                ops.append((rng.choice("IIDDS"), rng.choice([1, 2, 3, 4, 9])))
            span = sum(l for o, l in ops if o in "MD")
            pos = start - 10 - rng.randint(0, max(1, span // 3))
            idx.append(bb.add_read(pos, B.encode_cigar(ops), phase=1 + (k & 1), is_2d=(k % 5 == 0), mapq=rng.choice([9, 60, 60])))
        bb.add_locus(start, end, idx)
    batch = bb.build()
    rc, got = ctx.call_batch(batch, debug=True)
    oc, want = orc.call_batch(batch, debug=True)
    assert rc == oc == 0
    assert (np.abs(want.pair_call) > 100).sum() > 10
    _assert_same(got, want, f"wide unphased={unphased}")


def test_empty_and_ragged(ctx, orc):
    rc, res = ctx.call_batch(B.BatchBuilder().build())
    assert rc == 0 and res.phase1.shape == (0,)
    bb = B.BatchBuilder(support=1)
    bb.add_locus(100, 200, [])
    r0 = bb.add_read(50, np.zeros(0, dtype=np.uint32), phase=1)  # empty CIGAR: endpos = pos + 1
    r1 = bb.add_read(150, B.encode_cigar([("M", 1)]), phase=2)
    bb.add_locus(100, 200, [r0, r1])
    bb.add_locus(300, 300, [r1])
    batch = bb.build()
    rc, got = ctx.call_batch(batch, debug=True)
    oc, want = orc.call_batch(batch, debug=True)
    assert rc == oc == 0
    _assert_same(got, want)
    assert np.isnan(got.phase1[0]) and np.isnan(got.phase2[0])


def test_long_cigars(ctx, orc):
    """Reads of 1 .. 70 000 ops (beyond the 65 535 a BAM record stores inline), chunk edges included."""
    import random

    rng = random.Random(3)
    bb = B.BatchBuilder(support=1)
    for n_ops in (1, 255, 256, 257, 511, 512, 513, 1024, 4097, 70_000):
        cig = gen.random_cigar(rng, n_ops)
        rlen = sum(l for o, l in cig if o in "MDN=X")
        start = 1_000_000 + rlen // 2
        idx = [bb.add_read(1_000_000, B.encode_cigar(cig), phase=1 + (k & 1)) for k in range(3)]
        bb.add_locus(start, start + 200, idx)
    batch = bb.build()
    rc, got = ctx.call_batch(batch, debug=True)
    oc, want = orc.call_batch(batch, debug=True)
    assert rc == oc == 0
    _assert_same(got, want)


@pytest.mark.parametrize("unphased,support", [(False, 3), (True, 3), (False, 15_000), (True, 10_000)])
def test_one_locus_of_100000_reads(ctx, orc, unphased, support):
    """Amplicon depth: one locus offered 100 000 reads (plus two ordinary loci in the same batch).  The walk of such a locus is
    spread over the whole grid, the reduce is a radix select over the per-read Calls in global memory; exact against the
    oracle, per-pair outputs included.  support = 15 000 / 10 000 makes `spanning <= support` true at depth: the clip rule of
    median_str_length (src/call.rs:509-513) then picks tens of thousands of soft-clipped Calls by value."""
    import random
    import time

    rng = random.Random(77)
    start, end = 500_000, 500_120
    shapes = gen.random_locus_reads(rng, start, end, 300, long_every=13)
    bb = B.BatchBuilder(minlen=5, support=support, unphased=unphased)
    deep = []
    for k in range(100_000):
        r = shapes[rng.randrange(len(shapes))]
        deep.append(bb.add_read(r.pos, B.encode_cigar(r.cigar), mapq=rng.choice([9, 30, 60, 60, 60]), phase=rng.choice([None, 0, 1, 1, 2, 2]),
                                reverse=bool(r.flag & 0x10), is_2d=(k % 11 == 0)))
    order = sorted(range(len(deep)), key=lambda i: (bb._reads[deep[i]][2], i))  # file order: by position, then insertion
    bb.add_locus(start - 4000, start - 3900, deep[:40])
    bb.add_locus(start, end, [deep[i] for i in order])
    bb.add_locus(start + 9000, start + 9050, deep[100:130])
    batch = bb.build()
    t0 = time.perf_counter()
    rc, got = ctx.call_batch(batch, debug=True)
    dt = time.perf_counter() - t0
    oc, want = orc.call_batch(batch, debug=True, threads=8)
    assert rc == oc == 0
    _assert_same(got, want, f"100k reads unphased={unphased} support={support}")
    assert not np.isnan(got.phase2[1]), "the deep locus must yield a number for the test to mean anything"
    assert dt < 2.0, f"{dt:.2f} s for the host-buffer call (upload + kernels + download)"


_MILLION = {}


@pytest.mark.parametrize("unphased,support", [(False, 3), (True, 3), (False, 100_000), (True, 100_000)])
def test_one_locus_of_a_million_reads(ctx, orc, unphased, support):
    """One locus offered 1 000 000 reads, next to a 70 000-read one and two ordinary loci: beyond 65 536 reads the reduce is no
    longer one workgroup's (csrc/deep_select.hip: every pass of the radix selects is a launch over the whole grid; the unphased
    split's file-order tie rule a prefix over per-slice counts).  Exact against the oracle, per-pair outputs included; support =
    100 000 makes `spanning <= support` true at this depth, so the clip rule of median_str_length (src/call.rs:509-513) picks
    soft-clipped Calls by value, ties at the threshold included.  The whole launch sequence (walk + every reduce) is timed."""
    import random

    if "reads" not in _MILLION:  # the same reads for the four cases: only the scalars of the batch differ
        rng = random.Random(1234)
        start, end = 700_000, 700_150
        shapes = gen.random_locus_reads(rng, start, end, 400, long_every=17)
        _MILLION["reads"] = [(shapes[rng.randrange(len(shapes))], rng.choice([9, 30, 60, 60, 60]), rng.choice([None, 0, 1, 1, 2, 2]), k % 13 == 0)
                             for k in range(1_000_000)]
        _MILLION["win"] = (start, end)
    start, end = _MILLION["win"]
    bb = B.BatchBuilder(minlen=5, support=support, unphased=unphased)
    deep = [bb.add_read(r.pos, B.encode_cigar(r.cigar), mapq=mq, phase=ph, reverse=bool(r.flag & 0x10), is_2d=twod) for r, mq, ph, twod in _MILLION["reads"]]
    order = sorted(range(len(deep)), key=lambda i: (bb._reads[deep[i]][2], i))  # file order: by position, then insertion
    bb.add_locus(start - 4000, start - 3900, deep[:40])
    bb.add_locus(start, end, [deep[i] for i in order])
    bb.add_locus(start + 20, end + 20, [deep[i] for i in order[:70_000]])
    bb.add_locus(start + 9000, start + 9050, deep[100:130])
    batch = bb.build()
    ctx.call_batch(batch)  # allocations
    ctx.timing_enable(True)
    ctx.timing_reset()
    rc, got = ctx.call_batch(batch, debug=True)
    seq_ms, launches = ctx.timing_read(0)
    ctx.timing_enable(False)
    oc, want = orc.call_batch(batch, debug=True, threads=8)
    assert rc == oc == 0
    _assert_same(got, want, f"a million reads unphased={unphased} support={support}")
    assert not np.isnan(got.phase2[1]), "the deepest locus must yield a number for the test to mean anything"
    assert np.isnan(got.phase2[2]) == (support > 3)  # the 70 000-read locus: a number, or (fewer Calls per group than `support`) NaN
    print(f"1 000 000-read locus, unphased={unphased} support={support}: launch sequence {seq_ms:.2f} ms")
    assert launches == 1 and seq_ms < 5.0, f"{seq_ms:.2f} ms for walk + reduce of the batch"


@pytest.mark.parametrize("seed,unphased", [(1, False), (2, True), (3, False), (4, True)])
def test_very_deep_loci_at_the_edges_of_the_clip_rule(ctx, orc, seed, unphased):
    """Three loci of 66 000 - 80 000 reads (the grid-wide select of csrc/deep_select.hip takes over above 65 536), values drawn
    from few shapes so that ties are everywhere, and `support` placed ON the edges of median_str_length's rule for the deepest
    locus' first haplotype (src/call.rs:497-513): spanning - 1 / spanning / spanning + 1 (does the clip rule apply, with how many
    clipped Calls), all Calls (every clipped one taken), one more (NaN).  Exact against the oracle each time."""
    import random

    rng = random.Random(9000 + seed)
    start, end = 900_000, 900_140
    shapes = gen.random_locus_reads(rng, start, end, 60, long_every=9)
    pool = [(shapes[rng.randrange(len(shapes))], rng.choice([9, 60, 60, 60]), rng.choice([None, 0, 1, 1, 2, 2]), k % 7 == 0) for k in range(80_000)]

    def build(support):
        bb = B.BatchBuilder(minlen=5, support=support, unphased=unphased)
        ids = [bb.add_read(r.pos, B.encode_cigar(r.cigar), mapq=mq, phase=ph, reverse=bool(r.flag & 0x10), is_2d=twod) for r, mq, ph, twod in pool]
        order = sorted(range(len(ids)), key=lambda i: (bb._reads[ids[i]][2], i))
        bb.add_locus(start, end, [ids[i] for i in order])
        bb.add_locus(start + 10, end + 10, [ids[i] for i in order[:66_000]])
        bb.add_locus(start - 10, end - 10, [ids[i] for i in order[5_000:77_000]])
        bb.add_locus(start + 9000, start + 9050, ids[100:130])
        return bb.build()

    batch = build(1)
    oc, probe = orc.call_batch(batch, debug=True, threads=8)
    assert oc == 0
    n0 = int(batch.locus_pair_off[1])
    bits = probe.pair_bits[:n0].astype(np.int64)
    kept = (bits & 4) != 0
    if unphased:  # h1 = the lower half of the kept Calls
        vals = np.sort(probe.pair_call[:n0][kept], kind="stable")
        h1 = kept.sum() // 2
        ng = int(h1)
        ns = None  # which Calls are clipped depends on the split: take the supports around half of h1 instead
        supports = [1, max(1, ng // 2), ng - 1, ng, ng + 1]
        del vals
    else:
        phase = batch.reads["phase"][batch.pair_read[:n0]]
        g1 = kept & (phase == 1)
        ng, ns = int(g1.sum()), int((g1 & ((bits & 1) == 0)).sum())
        supports = [max(1, ns - 1), ns, ns + 1, ng, ng + 1]
    for support in supports:
        b = build(support)
        rc, got = ctx.call_batch(b, debug=True)
        oc, want = orc.call_batch(b, debug=True, threads=8)
        assert rc == oc == 0
        _assert_same(got, want, f"very deep loci seed={seed} unphased={unphased} support={support} (ng {ng}, ns {ns})")
    assert ng > 10_000


@pytest.mark.parametrize("unphased", [False, True])
@pytest.mark.parametrize("spread", ["one_value", "two_bytes", "five_bytes", "negative_and_huge"])
def test_very_deep_locus_with_calls_of_every_spread(ctx, orc, unphased, spread):
    """The grid-wide select (csrc/deep_select.hip) works on keys rebased to the locus' smallest Call and starts at the most
    significant byte their spread reaches: a 70 000-read locus whose Calls are all EQUAL (one pass over a single bin), spread over
    two bytes, over five (several 28-bit insertions per read: Calls beyond 2^32), and from large negative (deletions) to huge
    positive - next to a second very deep locus of ordinary spread, so that loci of different pass counts share the passes.
    Exact against the oracle, per-pair outputs included, for support 3 and for a support that makes the clip rule bite."""
    import random

    rng = random.Random({"one_value": 1, "two_bytes": 2, "five_bytes": 3, "negative_and_huge": 4}[spread])
    big = (1 << 28) - 1
    start, end = 800_000, 800_100

    def cigar():
        if spread == "one_value":
            return [("M", 150), ("I", 17), ("M", 200)]
        if spread == "two_bytes":
            return [("M", 150), ("I", rng.randint(6, 40_000)), ("M", 200)]
        if spread == "five_bytes":
            return [("M", 150)] + [x for _ in range(rng.choice([0, 1, 3, 9])) for x in (("I", rng.choice([big, 1 << 24, 77])), ("M", 1))] + [("M", 200)]
        if rng.random() < 0.5:  # a deletion inside the window, nothing else
            return [("M", 140), ("D", rng.choice([6, 2000, 60_000])), ("M", 300)]
        return [("M", 150)] + [x for _ in range(rng.choice([1, 5])) for x in (("I", big), ("M", 1))] + [("M", 200)]

    for support in (3, 20_000):
        bb = B.BatchBuilder(minlen=5, support=support, unphased=unphased)
        ids = []
        for k in range(70_000):
            cig = cigar()
            if k % 9 == 0:
                cig = [("S", 30)] + cig  # a soft clip that counts: the read starts inside the window (phased mode keeps it)
            pos = start - 10 - 100 if cig[0][0] != "S" else start + 5
            ids.append(bb.add_read(pos, B.encode_cigar(cig), mapq=60, phase=rng.choice([1, 2, 2]), is_2d=False))
        ordinary = gen.random_locus_reads(rng, start + 3000, end + 3000, 200, long_every=11)
        ids2 = [bb.add_read(r.pos, B.encode_cigar(r.cigar), mapq=60, phase=rng.choice([1, 2]), reverse=bool(r.flag & 0x10))
                for r in (ordinary[rng.randrange(len(ordinary))] for _ in range(66_500))]
        order = sorted(range(len(ids)), key=lambda i: (bb._reads[ids[i]][2], i))
        order2 = sorted(range(len(ids2)), key=lambda i: (bb._reads[ids2[i]][2], i))
        bb.add_locus(start, end, [ids[i] for i in order])
        bb.add_locus(start + 3000, end + 3000, [ids2[i] for i in order2])
        bb.add_locus(start + 9000, start + 9050, ids[100:130])
        batch = bb.build()
        rc, got = ctx.call_batch(batch, debug=True)
        oc, want = orc.call_batch(batch, debug=True, threads=8)
        assert rc == oc == 0
        _assert_same(got, want, f"spread={spread} unphased={unphased} support={support}")
        if spread == "five_bytes":
            assert np.abs(want.pair_call).max() > (1 << 31)
        if support == 3:
            assert not np.isnan(got.phase2[0]) and not np.isnan(got.phase2[1])


@pytest.mark.parametrize("case", range(8))
def test_batches_that_mix_every_depth_class(ctx, orc, case):
    """One batch with loci of every depth class side by side (gen.mixed_depth_case: <= 64, 65 - 256, 257 - 2 048, 2 049 - 16 384, several
    of 16 385 - 65 536, cases 0 and 4 one beyond 65 536; class boundaries themselves; Calls of few or many distinct values; support 1 ...
    9 000; both modes; with and without the caller's depth hint): the three launches behind one call share work lists, scratch and the
    persistent tail - rows, per-pair Calls and bits, tie counts equal the oracle's.  tools/soak_deep.py runs hundreds of these."""
    batch, depths = gen.mixed_depth_case(880_000 + case, case)
    ctx.set_option("max_reads_hint", 0 if case % 3 else int(max(depths)))
    try:
        rc, got = ctx.call_batch(batch, debug=True)
    finally:
        ctx.set_option("max_reads_hint", 0)
    oc, want = orc.call_batch(batch, debug=True, threads=8)
    assert rc == oc == 0
    _assert_same(got, want, f"mixed depths case {case}: {sorted(depths)[-4:]}")
    assert max(depths) > (65_536 if case % 4 == 0 else 16_384) and batch.n_loci == len(depths)


def test_allocation_that_meets_out_of_memory_frees_the_parked_buffers_and_tries_again(orc):
    """ensure() parks a buffer it has outgrown instead of freeing it (no wait for the device); an allocation that then fails for lack
    of memory must give those back and try again rather than fail the call.  The failure is injected ("test_fail_allocs":
    the next N allocations see out-of-memory at their first attempt); growing batches make the context outgrow and park its buffers;
    every call must still return the oracle's rows, the retries must have happened, and with "retired_limit_mb" = 0 nothing stays parked."""
    from inquistr_amd import hipcall

    with hipcall.Context(0) as c:
        wl = synth.WORKLOADS["phased10k"]
        for round_, (n_loci, fail) in enumerate(((64, 0), (700, 5), (3000, 50), (3001, 0))):
            batch = synth.generate_numpy(wl, 0, n_loci)
            if round_ == 2:
                c.set_option("retired_limit_mb", 0)
            c.set_option("test_fail_allocs", fail)
            before = c.alloc_retries
            rc, got = c.call_batch(batch, debug=True)
            oc, want = orc.call_batch(batch, debug=True)
            assert rc == oc == 0
            _assert_same(got, want, f"round {round_}")
            assert c.alloc_retries - before >= min(fail, 1) * 2  # the injected failure and the second attempt
        c.set_option("test_fail_allocs", 0)


def test_domain_errors(ctx, orc):
    def one(**kw):
        bb = B.BatchBuilder(**{k: v for k, v in kw.items() if k in ("minlen", "support", "unphased")})
        r = bb.add_read(pos=kw.get("pos", 900), cigar_words=kw.get("cigar", B.encode_cigar([("M", 300)])),
                        phase=kw.get("phase", 1))
        bb.add_locus(kw.get("start", 1010), kw.get("end", 1090), [r])
        return bb.build()

    def both(b):
        rc, _ = ctx.call_batch(b, check=False)
        assert rc == orc.call_batch(b)[0]
        return rc

    assert both(one(support=0)) == B.INQ_ERR_SUPPORT_ZERO
    assert both(one(start=9, end=90)) == B.INQ_ERR_LOCUS
    assert both(one(start=100, end=99)) == B.INQ_ERR_LOCUS
    assert both(one(phase=3)) == B.INQ_ERR_PHASE
    assert both(one(phase=3, unphased=True)) == B.INQ_OK
    assert both(one(cigar=np.array([(300 << 4) | 9], dtype=np.uint32))) == B.INQ_ERR_CIGAR_OP
    assert both(one(pos=2**31 - 200)) == B.INQ_ERR_RANGE
    b = one()
    b.pair_read[0] = 5
    assert both(b) == B.INQ_ERR_INDEX
    b = one()
    b.reads["n_cigar"][0] = 9
    assert both(b) == B.INQ_ERR_INDEX
    b = one()
    b.locus_pair_off[1] = 2
    assert both(b) == B.INQ_ERR_ARG
    # INQ_READ_SA_PANIC: raised for a KEPT read only (src/call.rs:303,357 -> :394); the read below has a soft clip
    def sa(mapq, phase, unphased, pos=900, depth=1):
        bb = B.BatchBuilder(unphased=unphased)
        good = [bb.add_read(900, B.encode_cigar([("M", 300)]), phase=1) for _ in range(depth - 1)]
        r = bb.add_read(pos, B.encode_cigar([("S", 20), ("M", 300)]), mapq=mapq, phase=phase, sa_panic=True)
        bb.add_locus(1010, 1090, good + [r])
        return bb.build()

    for depth in (1, 70, 300, 3000):  # every locus kernel shares the read epilogue
        assert both(sa(60, 1, False, depth=depth)) == B.INQ_ERR_AUX
        assert both(sa(60, 1, True, depth=depth)) == B.INQ_ERR_AUX
        assert both(sa(10, 1, False, depth=depth)) == B.INQ_OK      # mapq <= 10: filtered
        assert both(sa(60, None, False, depth=depth)) == B.INQ_OK   # no HP in phased mode: filtered
        assert both(sa(60, None, True, depth=depth)) == B.INQ_ERR_AUX
        assert both(sa(60, 1, True, pos=1005, depth=depth)) == B.INQ_OK   # starts inside the window: unphased drops it
        assert both(sa(60, 1, False, pos=1005, depth=depth)) == B.INQ_ERR_AUX  # phased keeps a partial overlap
        assert both(sa(60, 1, False, pos=5000, depth=depth)) == B.INQ_OK  # not fetched at all
    # the ctx stays usable after an error
    assert both(one()) == B.INQ_OK


@pytest.mark.parametrize("name", ["phased10k", "unphased100k", "expansion50k"])
def test_synthetic_workload_sample(ctx, orc, name):
    wl = synth.WORKLOADS[name]
    hi = 96 if wl.heavy_pct else 2000
    batch = synth.generate_numpy(wl, 0, hi)
    rc, got = ctx.call_batch(batch, debug=True)
    oc, want = orc.call_batch(batch, debug=True, threads=8)
    assert rc == oc == 0
    _assert_same(got, want, name)


def test_golden_random_fixture(ctx):
    """Committed vectors (tests/golden/random_batches.npz, made by tests/golden/make_fixtures.py
    from the CPU oracle): the GPU box needs neither the reference nor a rebuild to check them."""
    p = os.path.join(os.path.dirname(__file__), "golden", "random_batches.npz")
    z = np.load(p)
    n = int(z["n_cases"])
    for i in range(n):
        batch = B.Batch(
            cigar=z[f"c{i}_cigar"], reads=z[f"c{i}_reads"].view(B.READ_DTYPE).reshape(-1),
            pair_read=z[f"c{i}_pair_read"], locus_pair_off=z[f"c{i}_off"], locus_start=z[f"c{i}_start"],
            locus_end=z[f"c{i}_end"], minlen=int(z[f"c{i}_params"][0]), support=int(z[f"c{i}_params"][1]),
            unphased=bool(z[f"c{i}_params"][2]),
        )
        rc, got = ctx.call_batch(batch, debug=True)
        assert rc == 0
        want = B.Result(phase1=z[f"c{i}_p1"], phase2=z[f"c{i}_p2"], pair_call=z[f"c{i}_pair_call"],
                        pair_bits=z[f"c{i}_pair_bits"], n_tie_loci=int(z[f"c{i}_params"][3]))
        _assert_same(got, want, f"fixture case {i}")


def test_full_size_roofline_workload(ctx, orc):
    """BASELINE config #3 at full size, device-resident (100k loci x 30 reads x ~200 ops, unphased):
    exact against the oracle on three sampled locus ranges, plus size-independent properties —
    idempotence, and invariance of every locus' result to which other loci share the batch."""
    import torch

    wl = synth.WORKLOADS["unphased100k"]
    dev = torch.device("cuda:0")
    full = synth.DeviceBatch(wl, dev, debug=False)
    stream = torch.cuda.current_stream().cuda_stream
    ctx.call_batch_device(full.c_batch, full.c_result, stream)
    rc, ties = ctx.status()
    assert rc == 0
    p1, p2 = full.phase1.cpu().numpy(), full.phase2.cpu().numpy()
    # idempotence
    full.phase1.fill_(123.0)
    ctx.call_batch_device(full.c_batch, full.c_result, stream)
    rc, ties2 = ctx.status()
    assert rc == 0 and ties2 == ties
    assert gen.same_f64(full.phase1.cpu().numpy(), p1) and gen.same_f64(full.phase2.cpu().numpy(), p2)
    # exact on sampled ranges (the generator is counter based: numpy and torch agree per locus)
    for lo, hi in ((0, 400), (50_000, 50_400), (99_600, 100_000)):
        sub = synth.generate_numpy(wl, lo, hi)
        oc, want = orc.call_batch(sub, threads=8)
        assert oc == 0
        assert gen.same_f64(p1[lo:hi], want.phase1) and gen.same_f64(p2[lo:hi], want.phase2), (lo, hi)
    # every locus got a finite call in this workload and most medians are non-zero integers or halves
    assert not np.isnan(p1).any() and not np.isnan(p2).any()
    assert np.all(np.mod(p1 * 2, 1) == 0) and np.all(np.mod(p2 * 2, 1) == 0)
    assert np.all(p1 <= p2)  # unphased: h1 is the lower half of the sorted calls
    # a sub-batch resident on the device gives the same rows as the full batch
    part = synth.DeviceBatch(wl, dev, 70_000, 71_000)
    ctx.call_batch_device(part.c_batch, part.c_result, stream)
    assert ctx.status()[0] == 0
    assert gen.same_f64(part.phase1.cpu().numpy(), p1[70_000:71_000])
    assert gen.same_f64(part.phase2.cpu().numpy(), p2[70_000:71_000])


@pytest.mark.parametrize("name,lo,hi,samples", [
    ("phased10k", 0, 10_000, [(0, 300), (9_700, 10_000)]),                      # BASELINE config #2, whole
    ("expansion50k", 0, 50_000, [(0, 64), (25_000, 25_064), (49_936, 50_000)]),  # config #5, whole
    ("shard500k", 187_500, 250_000, [(187_500, 187_800), (249_700, 250_000)]),   # config #4: the shard of rank 3 of 8
])
def test_full_size_other_configs(ctx, orc, name, lo, hi, samples):
    """The other BASELINE configs at full size, device-resident: exact against the oracle on sampled locus
    ranges (the generator is counter based, so numpy reproduces any range), idempotent, no device error."""
    import torch

    wl = synth.WORKLOADS[name]
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    d = synth.DeviceBatch(wl, dev, lo, hi)
    ctx.call_batch_device(d.c_batch, d.c_result, st)
    rc, ties = ctx.status()
    assert rc == 0
    p1, p2 = d.phase1.cpu().numpy().copy(), d.phase2.cpu().numpy().copy()
    d.phase1.fill_(3.0)
    ctx.call_batch_device(d.c_batch, d.c_result, st)
    assert ctx.status() == (0, ties)
    assert gen.same_f64(d.phase1.cpu().numpy(), p1)
    for a, b in samples:
        sub = synth.generate_numpy(wl, a, b)
        oc, want = orc.call_batch(sub, threads=8)
        assert oc == 0
        assert gen.same_f64(p1[a - lo : b - lo], want.phase1) and gen.same_f64(p2[a - lo : b - lo], want.phase2), (name, a, b)
    assert np.all(np.mod(p1[~np.isnan(p1)] * 2, 1) == 0)


@pytest.mark.timeout(900)
def test_config4_all_500k_loci_as_eight_shards(ctx, orc):
    """BASELINE config #4 completely, on the one GPU there is: all 500 000 loci of shard500k (a) in one unsharded launch and
    (b) as the eight contiguous shards inquistr_amd.shard cuts them into (balanced by CIGAR-op count), every shard generated and
    called on its own as rank r would, the rows packed, and unpacked in rank order by the very code gather_rows() runs on either
    side of its collective (the collective itself: tests/test_shard_gloo.py, tests/test_call_dist_gloo.py at world 8).
    (b) must equal (a) bit for bit, and both the oracle on a sample from EVERY shard."""
    import torch

    from inquistr_amd import shard

    wl = synth.WORKLOADS["shard500k"]
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    full = synth.DeviceBatch(wl, dev, 0, wl.n_loci)
    ctx.call_batch_device(full.c_batch, full.c_result, st)
    rc, ties = ctx.status()
    assert rc == 0
    p1, p2 = full.phase1.cpu().numpy().copy(), full.phase2.cpu().numpy().copy()
    ops_per_locus = full.ops_per_locus().cpu().numpy()
    assert ops_per_locus.shape == (wl.n_loci,) and int(ops_per_locus.sum()) == full.n_ops_total
    del full
    world = 8
    ranges = shard.balanced_ranges(ops_per_locus, world)
    assert ranges[0][0] == 0 and ranges[-1][1] == wl.n_loci
    loads = [int(ops_per_locus[lo:hi].sum()) for lo, hi in ranges]
    assert max(loads) - min(loads) <= 2 * int(ops_per_locus.max())  # the cut is balanced by work, not by count
    width = max(hi - lo for lo, hi in ranges)
    bufs = []
    for r, (lo, hi) in enumerate(ranges):
        d = synth.DeviceBatch(wl, dev, lo, hi)
        ctx.call_batch_device(d.c_batch, d.c_result, st)
        assert ctx.status()[0] == 0, r
        bufs.append(shard.pack_rows(d.phase1, d.phase2, width))
        # the oracle on a sample of THIS shard: its first and last 150 loci
        for a, b in ((lo, lo + 150), (hi - 150, hi)):
            sub = synth.generate_numpy(wl, a, b)
            oc, want = orc.call_batch(sub, threads=8)
            assert oc == 0
            assert gen.same_f64(d.phase1[a - lo : b - lo].cpu().numpy(), want.phase1), (r, a, b)
            assert gen.same_f64(d.phase2[a - lo : b - lo].cpu().numpy(), want.phase2), (r, a, b)
        del d
    g1, g2 = shard.unpack_rows(bufs, ranges)
    assert g1.shape == (wl.n_loci,) and gen.same_f64(g1, p1) and gen.same_f64(g2, p2)
    assert not np.isnan(g1).any() and np.all(g1 <= g2)


def test_nccl_gather_is_available():
    """bench.py --gpus N moves the rows with ONE stated collective, dist.gather on the nccl (= RCCL) backend, and has no
    fallback: this exercises that very call on the GPU box (one rank: the backend's gather entry, buffers on the device)."""
    import socket

    import torch
    import torch.distributed as dist

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        src = torch.arange(8, dtype=torch.float64, device=dev).reshape(2, 4)
        out = [torch.empty_like(src)]
        work = dist.gather(src, out, dst=0, async_op=True)
        work.wait()
        torch.cuda.synchronize()
        assert torch.equal(out[0], src)
    finally:
        dist.destroy_process_group()


def test_bench_two_ranks_report_every_ranks_clock_and_the_gathers_tail():
    """bench.py --gpus 2 rehearsed on this box's one GPU (gloo for the gather, both ranks on device 0; with pytest's own process
    three processes on the card): the line carries every rank's own ms_per_step, the part of its gathers nothing overlapped, and its
    wait at the closing barrier - so that the first real scaling run explains itself - next to the contract's fields."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29653",
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--workload", "phased10k", "--backend", "gloo", "--same-device"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=root, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 6 and line["scaling"] == "weak" and line["value"] > 0
    assert [p["rank"] for p in line["per_rank"]] == [0, 1]
    for p in line["per_rank"]:
        assert 0 < p["kernels_ms_per_step"] <= p["ms_per_step"] <= line["ms_per_step"] * 1.0001
        assert p["gather_exposed_ms_total"] >= 0 and p["barrier_wait_ms"] >= 0
    assert line["gather"]["collectives"] == 2 and line["gather"]["bytes_per_rank_per_collective"] == 4 * 2 * 10_000 * 8


def test_bench_l2_dist_two_ranks_on_one_gpu():
    """bench.py --gpus 2 --l2-dist rehearsed on the one GPU (gloo, both ranks on device 0): the END-TO-END multi-rank mode - call_dist
    over one SEQ-bearing BAM, strong scaling - prints a line whose .inq equals the single-process CLI's and CPU mode B's byte for
    byte, with every rank's loci, BAM bytes, seconds and reader threads, the one-process `--devices 0,0` run beside it."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29657",
           os.path.join(root, "bench.py"), "--gpus", "2", "--l2-dist", "--l2-dist-loci", "1500", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--same-device"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=root, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert line["inq_identical_to_single_process"] is True
    assert line["cpu_baseline"]["inq_identical"] is True and line["cpu_baseline"]["kind"] == "port"
    assert line["native_devices"]["inq_identical"] is True and line["native_devices"]["devices"] == "0,0"
    assert [p["rank"] for p in line["per_rank"]] == [0, 1] and sum(p["loci"] for p in line["per_rank"]) == 1500
    read = [p["bam_bytes_read"] for p in line["per_rank"]]
    assert min(read) > 0.3 * line["config"]["bam_bytes"] and sum(read) < 1.15 * line["config"]["bam_bytes"]
    for p in line["per_rank"]:
        assert p["front"] == "device" and p["span_loop_GBps"] > 1 and p["io_threads"] >= 2 and p["io_threads"] * 2 <= max(p["granted_cpus"], 4)


def test_thousands_of_deep_loci_are_not_a_scan_of_the_work_list_per_workgroup(ctx):
    """3 000 loci of 270 and of 510 offered reads each (a targeted panel at several-hundred-fold depth: every locus on the deep work
    list).  Round 4 - and the first form of round 5's merged walk kernel - let every workgroup read through the whole list: 12 ms
    and 71 ms for 10 000 such loci.  A workgroup now takes its own items only and reduces a locus of up to 2 048 reads where it
    walked it: the rows must equal the plain batch's (every locus sees a superset of its reads) and the sequence must take well
    under a millisecond per thousand loci."""
    import torch

    wl = synth.WORKLOADS["phased10k"]
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream(device=dev)
    base = synth.DeviceBatch(wl, dev, 0, 3000)
    ctx.set_option("max_reads_hint", 0)
    ctx.call_batch_device(base.c_batch, base.c_result, stream.cuda_stream)
    torch.cuda.synchronize()
    assert ctx.status()[0] == 0
    p1, p2 = base.phase1.cpu().numpy(), base.phase2.cpu().numpy()
    for k in (4, 8):
        sup = synth.DeviceBatch(wl, dev, 0, 3000, neighbors=k)
        for _ in range(2):
            ctx.call_batch_device(sup.c_batch, sup.c_result, stream.cuda_stream)
        torch.cuda.synchronize()
        ctx.timing_enable(True)
        ctx.timing_reset()
        ctx.call_batch_device(sup.c_batch, sup.c_result, stream.cuda_stream)
        torch.cuda.synchronize()
        seq_ms, _ = ctx.timing_read(0)
        ctx.timing_enable(False)
        assert ctx.status()[0] == 0
        assert gen.same_f64(sup.phase1.cpu().numpy(), p1) and gen.same_f64(sup.phase2.cpu().numpy(), p2), k
        assert seq_ms < 1.5, f"{seq_ms:.2f} ms for 3 000 loci of {(2 * k + 1) * 30} reads"


def test_superset_of_candidates_changes_nothing(ctx):
    """The ABI lets the host offer more reads than fetch() would yield (one sweep over the file instead
    of an index query per locus): the device applies htslib's overlap rule.  Offering every locus the
    reads of its 1 / 2 neighbours on both sides (90 / 150 reads: the deep-locus kernel) must reproduce
    the exact rows of the plain batch, with every read now shared by 3 / 5 loci."""
    import torch

    wl = synth.WORKLOADS["phased10k"]
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    base = synth.DeviceBatch(wl, dev, 0, 3000)
    ctx.call_batch_device(base.c_batch, base.c_result, st)
    assert ctx.status()[0] == 0
    p1, p2 = base.phase1.cpu().numpy(), base.phase2.cpu().numpy()
    for k in (1, 2):
        sup = synth.DeviceBatch(wl, dev, 0, 3000, neighbors=k)
        assert sup.n_pairs > (2 * k) * base.n_pairs
        ctx.call_batch_device(sup.c_batch, sup.c_result, st)
        assert ctx.status()[0] == 0
        assert gen.same_f64(sup.phase1.cpu().numpy(), p1) and gen.same_f64(sup.phase2.cpu().numpy(), p2)
    # 1 neighbour on each side of 20 reads/locus keeps every locus at <= 60 reads: the wave-per-locus kernel
    wl20 = synth.Workload("p20", 3000, reads_per_locus=20, seed=9)
    b20 = synth.DeviceBatch(wl20, dev, 0, 3000)
    s20 = synth.DeviceBatch(wl20, dev, 0, 3000, neighbors=1)
    for d in (b20, s20):
        ctx.call_batch_device(d.c_batch, d.c_result, st)
    assert ctx.status()[0] == 0
    assert gen.same_f64(b20.phase1.cpu().numpy(), s20.phase1.cpu().numpy())
    assert gen.same_f64(b20.phase2.cpu().numpy(), s20.phase2.cpu().numpy())


def test_max_reads_hint(ctx):
    """A promise of <= 64 reads per locus skips the deep-locus launches; breaking it is reported."""
    import torch

    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    wl = synth.Workload("p70", 500, reads_per_locus=70, seed=3)
    d = synth.DeviceBatch(wl, dev, 0, 500)
    ctx.call_batch_device(d.c_batch, d.c_result, st)
    assert ctx.status()[0] == 0
    want = d.phase1.cpu().numpy().copy()
    try:
        ctx.set_option("max_reads_hint", 64)
        ctx.call_batch_device(d.c_batch, d.c_result, st)
        assert ctx.status()[0] == B.INQ_ERR_ARG
        for hint in (70, 256, 300, 2048, 5000, 9000, 70):  # every launch-skipping level, twice through the parities
            ctx.set_option("max_reads_hint", hint)
            d.phase1.fill_(7.0)
            ctx.call_batch_device(d.c_batch, d.c_result, st)
            assert ctx.status()[0] == 0 and gen.same_f64(d.phase1.cpu().numpy(), want), hint
        deep = synth.DeviceBatch(synth.Workload("p300", 200, reads_per_locus=300, seed=4), dev, 0, 200)
        ctx.set_option("max_reads_hint", 0)
        ctx.call_batch_device(deep.c_batch, deep.c_result, st)
        assert ctx.status()[0] == 0
        want_deep = deep.phase1.cpu().numpy().copy()
        for hint in (300, 2048, 2049, 8193, 256):
            ctx.set_option("max_reads_hint", hint)
            deep.phase1.fill_(7.0)
            ctx.call_batch_device(deep.c_batch, deep.c_result, st)
            rc = ctx.status()[0]
            if hint < 300:
                assert rc == B.INQ_ERR_ARG
            else:
                assert rc == 0 and gen.same_f64(deep.phase1.cpu().numpy(), want_deep), hint
    finally:
        ctx.set_option("max_reads_hint", 0)


def test_hip_graph_replay(ctx):
    """The device entry only enqueues: captured into a hipGraph (torch.cuda.CUDAGraph) and replayed, every
    replay must reproduce the rows — shallow loci, 65..256-read loci and deeper ones in one batch, so the
    work lists are filled and emptied on every replay."""
    import torch

    dev = torch.device("cuda:0")
    wl = synth.Workload("mix", 600, reads_per_locus=30, seed=8)
    d = synth.DeviceBatch(wl, dev, 0, 600, neighbors=6)  # 210 .. 390 offered reads per locus, edges fewer
    shallow = synth.DeviceBatch(synth.WORKLOADS["phased10k"], dev, 0, 600)
    st = torch.cuda.current_stream().cuda_stream
    for b in (d, shallow):  # eager reference + scratch allocation outside the capture
        ctx.call_batch_device(b.c_batch, b.c_result, st)
    assert ctx.status()[0] == 0
    want_d, want_s = d.phase1.cpu().numpy().copy(), shallow.phase1.cpu().numpy().copy()
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        with torch.cuda.graph(g, stream=cap):
            ctx.call_batch_device(d.c_batch, d.c_result, cap.cuda_stream)
            ctx.call_batch_device(shallow.c_batch, shallow.c_result, cap.cuda_stream)
    for _ in range(3):
        d.phase1.fill_(1.0)
        shallow.phase1.fill_(1.0)
        g.replay()
        torch.cuda.synchronize()
        assert gen.same_f64(d.phase1.cpu().numpy(), want_d) and gen.same_f64(shallow.phase1.cpu().numpy(), want_s)
    assert ctx.status()[0] == 0


def test_timing_events(ctx):
    import torch

    wl = synth.WORKLOADS["phased10k"]
    d = synth.DeviceBatch(wl, torch.device("cuda:0"), 0, 2000)
    ctx.timing_enable(True)
    ctx.timing_reset()
    for _ in range(3):
        ctx.call_batch_device(d.c_batch, d.c_result, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ms_all, n = ctx.timing_read(0)
    ms_k, n2 = ctx.timing_read(1)
    ctx.timing_enable(False)
    ctx.timing_reset()
    assert n == n2 == 3 and 0 < ms_k <= ms_all
    assert ctx.status()[0] == 0
