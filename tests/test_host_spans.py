"""Host half of the device front end (inq_spans_*): span planning, block tables and .bai anchors,
checked on the CPU by inflating the spans with zlib and replaying what the device kernels do."""
import numpy as np
import pytest

from inquistr_amd import batch as B
from inquistr_amd import call
from oracle import pyoracle as py
from tests import gen
from tests.test_host_frontend import _expected, _make_case
from tools import bamio


def emulate_span(sp):
    """The batch the device builds from one span, in plain Python: zlib inflate, record chain from the
    first anchor, htslib's overlap rule per locus."""
    u = bamio.inflate_span(sp["comp"], sp["blocks"])
    anchors = [int(a) for a in sp["anchors"]]
    recs = list(bamio.read_records(u, anchors[0]))
    starts = {r["off"] for r in recs}
    cut = recs[-1]["off"] if recs else 0
    for a in anchors:  # every anchor the chain reaches must be a record start
        assert a in starts or a > cut, a
    reads = [r for r in recs if r["tid"] == sp["tid"]]
    bb = B.BatchBuilder(minlen=sp["minlen"], support=sp["support"], unphased=sp["unphased"])
    ends = []
    for r in reads:
        ops = [("MIDNSHP=X"[w & 15], w >> 4) for w in r["cigar"]]
        rec = py.Record(pos=r["pos"], cigar=ops, mapq=r["mapq"], flag=r["flag"], hp=r["hp"], sa=r["sa"])
        ends.append(py.reference_end(rec))
        hp = r["hp"]
        has_clip = any(o == "S" for o, _ in ops)
        bb.add_read(r["pos"], np.array(r["cigar"], dtype=np.uint32), mapq=r["mapq"],
                    phase=(hp[1] & 0xFF) if hp and hp[0] in "Ci" else None,
                    reverse=bool(r["flag"] & 0x10), unmapped=bool(r["flag"] & 0x4),
                    is_2d=bool(has_clip and r["sa"] and py.is_accidental_2d(rec)))
    for s, e in zip(sp["locus_start"], sp["locus_end"]):
        lo, hi = int(s) - 10, int(e) + 10
        bb.add_locus(int(s), int(e), [i for i, r in enumerate(reads) if r["pos"] < hi and ends[i] > lo])
    return bb.build()


@pytest.mark.parametrize("seed,unphased,span_bytes", [(1, False, 0), (2, True, 20_000), (3, False, 3_000), (4, True, 1)])
def test_spans_reproduce_per_locus_fetch(tmp_path, orc, seed, unphased, span_bytes):
    minlen, support = 5, [3, 1, 2, 3][seed % 4]
    bam, bed, loci, recs = _make_case(tmp_path, seed, ultra_long=(seed == 3))
    sp = call.Spans(bam, region_file=bed, minlen=minlen, support=support, threads=3, unphased=unphased, max_comp_bytes=span_bytes)
    assert sp.n_targets == len(loci)
    got1 = np.full(len(loci), np.nan)
    got2 = np.full(len(loci), np.nan)
    seen = np.zeros(len(loci), dtype=int)
    n_spans = 0
    for span in sp.spans():
        n_spans += 1
        # whole blocks, dense output offsets, ascending anchors inside the inflated bytes
        blocks = span["blocks"]
        assert (blocks["out_off"][1:] == blocks["out_off"][:-1] + blocks["isize"][:-1]).all() and blocks["out_off"][0] == 0
        total = int(blocks["out_off"][-1] + blocks["isize"][-1])
        assert (np.diff(span["anchors"].astype(np.int64)) > 0).all() and int(span["anchors"][-1]) <= total
        batch = emulate_span(span)
        code, res = orc.call_batch(batch)
        assert code == 0
        idx = span["locus_index"]
        seen[idx] += 1
        got1[idx], got2[idx] = res.phase1, res.phase2
    assert seen.max() == 1  # a locus belongs to one span; loci in no span have no record near them
    if span_bytes and span_bytes < 50_000:
        assert n_spans > 3
    want1, want2 = _expected(loci, recs, unphased, minlen, support)
    assert gen.same_f64(got1, want1) and gen.same_f64(got2, want2)
    sp.close()


def test_spans_errors(tmp_path):
    with pytest.raises(call.CallError) as e:
        call.Spans(str(tmp_path / "missing.bam"), region="chr1:100-200")
    assert e.value.status == 1
