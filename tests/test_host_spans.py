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


SEG_END = 1 << 63


def emulate_span(sp):
    """The batch the device builds from one span, in plain Python: zlib inflate, record chains from the
    anchors of every segment, htslib's overlap rule per locus."""
    u = bamio.inflate_span(sp["comp"], sp["blocks"])
    anchors = [int(a) for a in sp["anchors"]]
    stops = [int(a) for a in sp["anchor_stop"]]
    recs = []
    for k, (a, st) in enumerate(zip(anchors, stops)):
        end = st & ~SEG_END
        chain = list(bamio.read_records(u[:end], a))
        nxt = chain[-1]["next"] if chain else a
        if st & SEG_END:
            assert nxt <= end
        else:
            assert nxt == end == anchors[k + 1], (k, nxt, end)  # a chain lands on the next anchor
        recs += chain
    reads = [r for r in recs if r["tid"] >= 0]
    keys = [(r["tid"], r["pos"]) for r in reads]
    assert keys == sorted(keys)
    bb = B.BatchBuilder(minlen=sp["minlen"], support=sp["support"], unphased=sp["unphased"])
    ends = []
    for r in reads:
        ops = [("MIDNSHP=X"[w & 15], w >> 4) for w in r["cigar"]]
        rec = py.Record(pos=r["pos"], cigar=ops, mapq=r["mapq"], flag=r["flag"], hp=r["hp"], sa=r["sa"])
        ends.append(py.reference_end(rec))
        hp = r["hp"]
        has_clip = any(o == "S" for o, _ in ops)
        is_2d = sa_panic = False
        if has_clip and r["sa"]:  # what cigar_gather / FrontEnd::add_read evaluate, panic carried as a bit
            try:
                is_2d = py.is_accidental_2d(rec)
            except py.ReferencePanic:
                sa_panic = True
        bb.add_read(r["pos"], np.array(r["cigar"], dtype=np.uint32), mapq=r["mapq"],
                    phase=(hp[1] & 0xFF) if hp and hp[0] in "Ci" else None,
                    reverse=bool(r["flag"] & 0x10), unmapped=bool(r["flag"] & 0x4),
                    is_2d=is_2d, sa_panic=sa_panic)
    for t, s, e in zip(sp["locus_tid"], sp["locus_start"], sp["locus_end"]):
        lo, hi = int(s) - 10, int(e) + 10
        bb.add_locus(int(s), int(e), [i for i, r in enumerate(reads) if r["tid"] == int(t) and r["pos"] < hi and ends[i] > lo])
    return bb.build()


@pytest.mark.parametrize("seed,unphased,span_bytes,gap", [(1, False, 0, None), (2, True, 20_000, None), (3, False, 3_000, None),
                                                          (4, True, 1, None), (5, False, 0, 0), (6, True, 30_000, 0)])
def test_spans_reproduce_per_locus_fetch(tmp_path, orc, monkeypatch, seed, unphased, span_bytes, gap):
    if gap is not None:
        monkeypatch.setenv("INQ_SPAN_GAP_BYTES", str(gap))  # every gap between loci opens a new segment
    minlen, support = 5, [3, 1, 2, 3][seed % 4]
    # small BGZF blocks when segments are wanted: the test file is far smaller than real gaps
    bam, bed, loci, recs = _make_case(tmp_path, seed, ultra_long=(seed == 3), block=bamio.BLOCK if gap is None else 1500)
    sp = call.Spans(bam, region_file=bed, minlen=minlen, support=support, threads=3, unphased=unphased, max_comp_bytes=span_bytes)
    assert sp.n_targets == len(loci)
    got1 = np.full(len(loci), np.nan)
    got2 = np.full(len(loci), np.nan)
    seen = np.zeros(len(loci), dtype=int)
    n_spans = n_segments = 0
    for span in sp.spans():
        n_spans += 1
        n_segments += int((span["anchor_stop"] >= np.uint64(SEG_END)).sum())
        # whole blocks, dense output offsets, ascending anchors inside the inflated bytes
        blocks = span["blocks"]
        assert (blocks["out_off"][1:] == blocks["out_off"][:-1] + blocks["isize"][:-1]).all() and blocks["out_off"][0] == 0
        total = int(blocks["out_off"][-1] + blocks["isize"][-1])
        assert (np.diff(span["anchors"].astype(np.int64)) > 0).all() and int(span["anchors"][-1]) <= total
        assert int(span["anchor_stop"][-1]) == total | SEG_END
        batch = emulate_span(span)
        code, res = orc.call_batch(batch)
        assert code == 0
        idx = span["locus_index"]
        seen[idx] += 1
        got1[idx], got2[idx] = res.phase1, res.phase2
    assert seen.max() == 1  # a locus belongs to one span; loci in no span have no record near them
    if span_bytes and span_bytes < 50_000:
        assert n_spans > 3
    if gap == 0 and not span_bytes:
        assert n_spans == 1 and n_segments > 1  # all contigs in one call, made of several pieces of the file
    want1, want2 = _expected(loci, recs, unphased, minlen, support)
    assert gen.same_f64(got1, want1) and gen.same_f64(got2, want2)
    sp.close()


def test_spans_errors(tmp_path):
    with pytest.raises(call.CallError) as e:
        call.Spans(str(tmp_path / "missing.bam"), region="chr1:100-200")
    assert e.value.status == 1


@pytest.mark.parametrize("threads,pin", [(1, 0), (2, 1), (5, 1), (16, 1), (16, 0), (33, 1)])
def test_reader_pool_runs_every_job_exactly_once(threads, pin):
    """The pool of reader threads the span loader copies file bytes with (made once per file, bound in turn to the L3 domains of
    a NUMA node where sysfs shows any): many rounds of few and of many jobs, fewer jobs than threads, none at all - every job runs
    exactly once per round, whatever the thread count and the binding (node 0, the machine, a node that does not exist)."""
    from inquistr_amd import call

    L = call.load()
    for node in (0, -1, 77):
        for n_jobs, rounds in ((0, 3), (1, 50), (3, 200), (64, 100), (1000, 20)):
            want = rounds * n_jobs * (n_jobs + 1) // 2
            assert L.inq_host_iopool_selftest(threads, node, pin, n_jobs, rounds) == want, (threads, pin, node, n_jobs, rounds)


@pytest.mark.parametrize("seed,block,gap", [(11, 1500, None), (12, 700, 0), (13, bamio.BLOCK, None)])
def test_block_tables_do_not_depend_on_how_the_copy_is_dealt(tmp_path, monkeypatch, seed, block, gap):
    """The span loader cuts a span's bytes into jobs at index anchors (every virtual offset of the index names a BGZF block start); the
    thread that copies a job's bytes also hops through its block headers.  Whatever the job size - one job per piece, a job per anchor,
    something between - the compressed bytes, the block table (payload offsets, lengths, ISIZE, dense output offsets) and the anchors
    are the same, and every block inflates (zlib) to its ISIZE."""
    import zlib

    if gap is not None:
        monkeypatch.setenv("INQ_SPAN_GAP_BYTES", str(gap))
    bam, bed, loci, recs = _make_case(tmp_path, seed, n_loci=90, ultra_long=True, block=block)

    def collect(job_bytes):
        if job_bytes is None:
            monkeypatch.delenv("INQ_SPAN_JOB_BYTES", raising=False)
        else:
            monkeypatch.setenv("INQ_SPAN_JOB_BYTES", str(job_bytes))
        sp = call.Spans(bam, region_file=bed, threads=4, max_comp_bytes=0)
        out = []
        for span in sp.spans():
            out.append((bytes(span["comp"]), span["blocks"].copy(), span["anchors"].copy(), span["anchor_stop"].copy(), span["locus_index"].copy()))
        sp.close()
        return out

    plain = collect(None)
    assert plain and sum(len(s[1]) for s in plain) > (20 if block < 5000 else 2)
    for job_bytes in (1, 4000, 50_000):
        got = collect(job_bytes)
        assert len(got) == len(plain)
        for a, b in zip(plain, got):
            assert a[0] == b[0]
            assert a[1].tobytes() == b[1].tobytes() and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
    for comp, blocks, *_ in plain:
        for blk in blocks[:: max(1, len(blocks) // 40)]:
            o, n = int(blk["comp_off"]), int(blk["comp_len"])
            assert len(zlib.decompressobj(-15).decompress(comp[o : o + n])) == int(blk["isize"])
