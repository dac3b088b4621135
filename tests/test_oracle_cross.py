"""C oracle (batch form over the C-ABI structs) vs the independent Python restatement."""
import numpy as np
import pytest

from tests import gen
from inquistr_amd import batch as B
from oracle import pyoracle as py


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("unphased", [False, True])
def test_batch_matches_python(orc, seed, unphased):
    support = [3, 1, 2, 5][seed % 4]
    minlen = [5, 0, 5, 12][seed % 4]
    batch, per_locus = gen.random_case(seed, n_loci=30, unphased=unphased, minlen=minlen, support=support,
                                       long_every=7 if seed % 3 == 0 else 0)
    p1, p2, ties = gen.py_expected(batch, per_locus)
    code, res = orc.call_batch(batch, debug=True)
    assert code == B.INQ_OK
    assert gen.same_f64(res.phase1, p1) and gen.same_f64(res.phase2, p2)
    if unphased:
        assert res.n_tie_loci == ties
    # per-pair debug outputs follow call_from_cigar for every offered read
    k = 0
    for j, recs in enumerate(per_locus):
        se, ee = int(batch.locus_start[j]) - 10, int(batch.locus_end[j]) + 10
        for r in recs:
            kind, val = py.call_from_cigar(r, batch.minlen, se, ee)
            assert res.pair_call[k] == val
            assert bool(res.pair_bits[k] & B.INQ_PAIR_CLIP) == (kind == "Clip")
            k += 1
    assert k == batch.n_pairs


def test_threads_do_not_change_results(orc):
    batch, _ = gen.random_case(99, n_loci=200, unphased=True)
    c1, r1 = orc.call_batch(batch, threads=1)
    c4, r4 = orc.call_batch(batch, threads=4)
    assert c1 == c4 == 0
    assert gen.same_f64(r1.phase1, r4.phase1) and gen.same_f64(r1.phase2, r4.phase2)
    assert r1.n_tie_loci == r4.n_tie_loci


def test_empty_batch(orc):
    bb = B.BatchBuilder()
    code, res = orc.call_batch(bb.build())
    assert code == 0 and res.phase1.shape == (0,)
    bb = B.BatchBuilder()
    bb.add_locus(100, 200, [])
    code, res = orc.call_batch(bb.build())
    assert code == 0 and np.isnan(res.phase1[0]) and np.isnan(res.phase2[0])


def test_domain_errors(orc):
    def one(**kw):
        bb = B.BatchBuilder(**{k: v for k, v in kw.items() if k in ("minlen", "support", "unphased")})
        r = bb.add_read(pos=kw.get("pos", 900), cigar_words=kw.get("cigar", B.encode_cigar([("M", 300)])),
                        phase=kw.get("phase", 1))
        bb.add_locus(kw.get("start", 1010), kw.get("end", 1090), [r])
        return bb.build()

    assert orc.call_batch(one(support=0))[0] == B.INQ_ERR_SUPPORT_ZERO
    assert orc.call_batch(one(start=9, end=90))[0] == B.INQ_ERR_LOCUS
    assert orc.call_batch(one(start=100, end=99))[0] == B.INQ_ERR_LOCUS
    assert orc.call_batch(one(phase=3))[0] == B.INQ_ERR_PHASE
    assert orc.call_batch(one(phase=3, unphased=True))[0] == B.INQ_OK
    assert orc.call_batch(one(cigar=np.array([(300 << 4) | 9], dtype=np.uint32)))[0] == B.INQ_ERR_CIGAR_OP
    assert orc.call_batch(one(pos=2**31 - 200))[0] == B.INQ_ERR_RANGE
    b = one()
    b.pair_read[0] = 5
    assert orc.call_batch(b)[0] == B.INQ_ERR_INDEX
    b = one()
    b.reads["n_cigar"][0] = 9
    assert orc.call_batch(b)[0] == B.INQ_ERR_INDEX
    b = one()
    b.locus_pair_off[1] = 2
    assert orc.call_batch(b)[0] == B.INQ_ERR_ARG
