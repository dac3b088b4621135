"""Device front end (inq_bgzf_inflate / inq_call_span) against zlib, the host front end and the oracle."""
import os
import random
import struct
import zlib

import numpy as np
import pytest

from inquistr_amd import hipcall

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[(0, 0, 1), (0, 1, 0), (0, -1, -1), (1, 0, -1)],
                ids=["inflate_wg_literal_pairs", "inflate_wg_no_pairs_commit_from_tokens", "inflate_wg_form_by_the_data", "inflate_lane"])
def ctx(request):
    """Both inflate kernels go through every test of this file: workgroup per block - its symbol loop with and without the second
    literal per peek, its commit decoding again and fed from tokens, and the form chosen from the block headers as the product
    does - and lane per block."""
    c = hipcall.Context(0)
    c.set_option("inflate_algo", request.param[0])
    c.set_option("inflate_tokens", request.param[1])
    c.set_option("inflate_lit_pairs", request.param[2])
    yield c
    c.close()


def _bgzf(data: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=-15) -> bytes:
    co = zlib.compressobj(level, zlib.DEFLATED, wbits, 8, strategy)
    payload = co.compress(data) + co.flush()
    bsize = 18 + len(payload) + 8
    assert bsize <= 65536
    hdr = b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize - 1)
    return hdr + payload + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))


def _payloads(rng: random.Random):
    """Byte strings that push zlib into every block type and every length / distance class."""
    out = []
    out.append(b"")
    out.append(b"a")
    out.append(bytes(65280))  # one symbol, distance-1 matches of length 258
    out.append(bytes(rng.getrandbits(8) for _ in range(60000)))  # incompressible: stored blocks
    out.append((b"ACGT" * 7 + b"N") * 2000)  # short period, long matches
    words = [bytes(rng.getrandbits(8) for _ in range(rng.randint(1, 40))) for _ in range(300)]
    out.append(b"".join(rng.choice(words) for _ in range(3000))[:65280])  # dictionary-like: all distances
    # BAM-like: little-endian integers with small deltas
    pos, recs = 1000, []
    for _ in range(1500):
        pos += rng.randint(0, 30)
        recs.append(struct.pack("<iiBBHHHI", 0, pos, 8, 60, 4681, rng.randint(1, 9), 16 * rng.randint(0, 1), 0))
        recs.append(b"".join(struct.pack("<I", rng.randint(1, 300) << 4 | rng.choice([0, 1, 2])) for _ in range(rng.randint(1, 9))))
    out.append(b"".join(recs)[:65280])
    far = bytes(rng.getrandbits(8) for _ in range(300))
    out.append(far + bytes(rng.getrandbits(8) for _ in range(32000)) + far)  # a match at distance > 32000
    return out


def test_inflate_matches_zlib(ctx):
    rng = random.Random(5)
    blobs, want = [], []
    for data in _payloads(rng):
        for level, strategy in [(0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY),
                                (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE),
                                (9, zlib.Z_FILTERED)]:
            if level == 0 and len(data) > 65000:
                data = data[:65000]  # stored blocks add 5 bytes per 64 KB
            try:
                blobs.append(_bgzf(data, level, strategy))
            except AssertionError:
                continue  # incompressible data that does not fit one block at this setting
            want.append(data)
    comp = b"".join(blobs)
    blocks = hipcall.scan_bgzf(comp)
    assert len(blocks) == len(want) and len(want) > 50
    rc, out, status = ctx.bgzf_inflate(comp, blocks)
    assert rc == 0 and not status.any()
    for b, w in zip(blocks, want):
        got = out[int(b["out_off"]) : int(b["out_off"]) + int(b["isize"])].tobytes()
        assert got == w, (len(w), w[:16])


def test_inflate_of_the_references_own_gzip_members(ctx):
    """The only compressed streams the reference repository holds (test-data/file{1,2,3}.inq.gz, written by another
    compressor than this repo's zlib): their DEFLATE payloads and CRC32 / ISIZE trailers through the device inflate."""
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    comp, table, want, out_off = b"", [], b"", 0
    for i in (1, 2, 3):
        gz = open(os.path.join(golden, f"reference_file{i}.inq.gz"), "rb").read()
        text = open(os.path.join(golden, f"reference_file{i}.inq"), "rb").read()
        assert gz[:3] == b"\x1f\x8b\x08" and gz[3] == 8  # FNAME only
        at = gz.index(b"\0", 10) + 1  # behind the zero-terminated file name
        crc, isize = struct.unpack("<II", gz[-8:])
        assert isize == len(text) and crc == zlib.crc32(text)
        # the kernel's unit is a BGZF block: 18 bytes of header in front of the payload, the trailer behind it
        comp += bytes(18) + gz[at:]
        table.append((len(comp) - len(gz[at:]), len(gz[at:]) - 8, isize, out_off))
        want += text
        out_off += isize
    blocks = np.array(table, dtype=hipcall.BGZF_BLOCK_DTYPE)
    rc, out, status = ctx.bgzf_inflate(comp, blocks)
    assert rc == 0 and not status.any()
    assert out.tobytes() == want


def test_inflate_multi_member_and_eof_block(ctx):
    # a deflate stream of several blocks inside one BGZF block (Z_FULL_FLUSH between them) + the EOF marker
    rng = random.Random(9)
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    parts = [bytes(rng.choice(b"ACGT") for _ in range(5000)), bytes(3000), bytes(rng.getrandbits(8) for _ in range(4000))]
    payload = b""
    for i, part in enumerate(parts):
        payload += co.compress(part) + co.flush(zlib.Z_FULL_FLUSH if i < 2 else zlib.Z_FINISH)
    data = b"".join(parts)
    bsize = 18 + len(payload) + 8
    blk = b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize - 1) + payload + \
        struct.pack("<II", zlib.crc32(data), len(data))
    eof = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    comp = blk + eof + blk
    blocks = hipcall.scan_bgzf(comp)
    assert [int(b["isize"]) for b in blocks] == [len(data), 0, len(data)]
    rc, out, status = ctx.bgzf_inflate(comp, blocks)
    assert rc == 0 and not status.any()
    assert out.tobytes() == data + data


def test_inflate_many_deflate_blocks_per_bgzf_block(ctx):
    """Hundreds of tiny deflate blocks (sync / full flushes, changing strategy) inside one BGZF block: every one has its own
    header, which the workgroup kernel decodes with all lanes."""
    rng = random.Random(77)
    blobs, want = [], []
    for case in range(24):
        n_cuts = rng.choice([5, 40, 150, 400])
        size = rng.choice([3000, 20000, 60000])
        kind = case % 3
        if kind == 0:
            data = bytes(rng.choice(b"ACGTN\x00\x10\x20") for _ in range(size))
        elif kind == 1:
            words = [bytes(rng.getrandbits(8) for _ in range(rng.randint(2, 30))) for _ in range(200)]
            data = b"".join(rng.choice(words) for _ in range(size // 10))[:size]
        else:
            data = bytes(rng.getrandbits(8) if rng.random() < 0.5 else 65 for _ in range(size))  # every byte value: full-size headers
        cuts = sorted(rng.randrange(0, len(data) + 1) for _ in range(n_cuts))
        co = zlib.compressobj(rng.choice([1, 6, 9]), zlib.DEFLATED, -15, 8, rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_RLE]))
        payload, prev = b"", 0
        for c in cuts:
            payload += co.compress(data[prev:c]) + co.flush(rng.choice([zlib.Z_FULL_FLUSH, zlib.Z_SYNC_FLUSH, zlib.Z_BLOCK]))
            prev = c
        payload += co.compress(data[prev:]) + co.flush()
        if 18 + len(payload) + 8 > 65536:
            continue
        hdr = b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", 18 + len(payload) + 8 - 1)
        blobs.append(hdr + payload + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))
        want.append(data)
    assert len(blobs) >= 12
    comp = b"".join(blobs)
    blocks = hipcall.scan_bgzf(comp)
    rc, out, status = ctx.bgzf_inflate(comp, blocks)
    assert rc == 0 and not status.any()
    assert out.tobytes() == b"".join(want)


def test_inflate_reports_corruption(ctx):
    rng = random.Random(11)
    data = bytes(rng.choice(b"ACGTN") for _ in range(30000))
    good = _bgzf(data)
    blocks = hipcall.scan_bgzf(good + good + good)
    comp = bytearray(good + good + good)
    # flip bytes inside the second block's payload; the neighbours must still come out right
    off = int(blocks[1]["comp_off"])
    for k in range(40, 60):
        comp[off + k] ^= 0x5A
    rc, out, status = ctx.bgzf_inflate(bytes(comp), blocks, check=False)
    n = len(data)
    assert out[:n].tobytes() == data and out[2 * n :].tobytes() == data
    assert status[0] == 0 and status[2] == 0
    # a flipped stream either fails a check or yields other bytes of the same length: zlib decides which
    try:
        alt = zlib.decompressobj(-15).decompress(bytes(comp[off : off + int(blocks[1]["comp_len"])]))
        zlib_ok = len(alt) == n
    except zlib.error:
        zlib_ok = False
    if zlib_ok:  # inflates, to other bytes: only the CRC32 of the trailer notices
        assert status[1] == 0x40 and out[n : 2 * n].tobytes() == alt and rc == hipcall.INQ_ERR_INFLATE
    else:
        assert status[1] != 0 and rc == hipcall.INQ_ERR_INFLATE
    # a wrong CRC32 in the trailer (htslib: read error); not looked at with verify_crc = 0
    bad_crc = bytearray(good + good + good)
    bad_crc[int(blocks[2]["comp_off"]) + int(blocks[2]["comp_len"])] ^= 1
    rc, out, status = ctx.bgzf_inflate(bytes(bad_crc), blocks, check=False)
    assert rc == hipcall.INQ_ERR_INFLATE and list(status) == [0, 0, 0x40] and out.tobytes() == data * 3
    ctx.set_option("verify_crc", 0)
    try:
        rc, out, status = ctx.bgzf_inflate(bytes(bad_crc), blocks, check=False)
        assert rc == 0 and not status.any()
    finally:
        ctx.set_option("verify_crc", 1)
    # wrong ISIZE
    blocks2 = blocks.copy()
    blocks2["isize"][1] -= 1
    blocks2["out_off"][2] -= 1
    rc, out, status = ctx.bgzf_inflate(good + good + good, blocks2, check=False)
    assert rc == hipcall.INQ_ERR_INFLATE and status[1] & 0x08 and status[0] == 0 and status[2] == 0
    # truncated payload
    blocks3 = blocks.copy()
    blocks3["comp_len"][0] -= 200
    rc, out, status = ctx.bgzf_inflate(good + good + good, blocks3, check=False)
    assert rc == hipcall.INQ_ERR_INFLATE and status[0] != 0 and status[1] == 0


# ---------------------------------------------------------------- spans: scan + join + call on the device
def _locus_view(batch_or_arrays, j):
    """Per-locus candidate list as comparable tuples (pos, mapq, bits, phase, cigar words)."""
    cigar, reads, pair_read, off = batch_or_arrays
    out = []
    for k in range(int(off[j]), int(off[j + 1])):
        r = reads[int(pair_read[k])]
        o = int(r["cigar_off4"]) * 4
        n = int(r["n_cigar"])
        assert not cigar[o + n : o + (n + 3) // 4 * 4].any()  # zero padding to 16 bytes
        out.append((int(r["pos"]), int(r["mapq"]), int(r["bits"]), int(r["phase"]) if r["bits"] & 4 else 0, cigar[o : o + n].tobytes()))
    return out


@pytest.mark.parametrize("seed,unphased,span_bytes,gap", [(1, False, 0, None), (2, True, 20_000, None), (3, False, 3_000, None),
                                                          (4, True, 1, None), (5, False, 0, 0), (6, True, 30_000, 0)])
def test_call_span_matches_oracle_and_host_emulation(ctx, tmp_path, monkeypatch, seed, unphased, span_bytes, gap):
    from inquistr_amd import call
    from tests import gen
    from tests.test_host_frontend import _expected, _make_case
    from tests.test_host_spans import emulate_span
    from tools import bamio

    if gap is not None:
        monkeypatch.setenv("INQ_SPAN_GAP_BYTES", str(gap))  # spans made of several segments of the file
    minlen, support = 5, [3, 1, 2, 3][seed % 4]
    bam, bed, loci, recs = _make_case(tmp_path, seed, ultra_long=(seed == 3), block=bamio.BLOCK if gap is None else 1500)
    sp = call.Spans(bam, region_file=bed, minlen=minlen, support=support, threads=2, unphased=unphased, max_comp_bytes=span_bytes)
    got1 = np.full(len(loci), np.nan)
    got2 = np.full(len(loci), np.nan)
    for k, span in enumerate(sp.spans()):
        # odd seeds go through inq_span_stage + inq_call_span_staged, rotating through the three device slots
        rc, p1, p2, ties, stats = ctx.call_span(span["comp"], span["blocks"], span["anchors"], span["anchor_stop"], span["locus_tid"],
                                                span["locus_start"], span["locus_end"], minlen, support, unphased,
                                                stage_slot=(k % 3) if seed % 2 else None)
        assert rc == 0
        idx = span["locus_index"]
        got1[idx], got2[idx] = p1, p2
        # the batch built on the device = the batch a plain-Python replay of the same span builds
        want = emulate_span(span)
        dev = ctx.span_fetch_batch(stats, len(idx))
        assert int(stats.n_pairs) == want.n_pairs and int(stats.n_reads) == want.n_reads
        ref = (want.cigar, want.reads, want.pair_read, want.locus_pair_off)
        for j in range(len(idx)):
            assert _locus_view(dev, j) == _locus_view(ref, j), (seed, j)
    want1, want2 = _expected(loci, recs, unphased, minlen, support)
    assert gen.same_f64(got1, want1) and gen.same_f64(got2, want2)
    sp.close()


@pytest.mark.parametrize("seed,unphased", [(21, False), (22, True)])
def test_two_spans_enqueued_before_the_first_is_waited_for(ctx, tmp_path, seed, unphased):
    """inq_span_stage_begin / _wait as the driver's uploader uses them (round 4): two spans are ENQUEUED (upload + the inflate behind
    it) into two of the eight slots before the first is waited for, so that the copy engine goes from one span's bytes straight to
    the next one's; the slots rotate, the tables travel through the slot's page-locked buffer (the caller's are scribbled over right
    after _begin), a slot that was begun and never waited for is usable again.  Rows = the Python restatement's."""
    from inquistr_amd import call
    from tests import gen
    from tests.test_host_frontend import _expected, _make_case

    minlen, support = 5, 2
    bam, bed, loci, recs = _make_case(tmp_path, seed, n_loci=70)
    sp = call.Spans(bam, region_file=bed, minlen=minlen, support=support, threads=2, unphased=unphased, max_comp_bytes=4_000)
    spans = []
    for span in sp.spans():  # (the iterator reuses its buffers: keep copies)
        spans.append({k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else bytes(v) if k == "comp" else v) for k, v in span.items()})
    sp.close()
    assert len(spans) >= 5
    # a slot begun and abandoned: the next _begin on it goes through
    s0 = spans[0]
    ctx.span_stage_begin(s0["comp"], s0["blocks"], s0["anchors"], s0["anchor_stop"], 7)
    got1 = np.full(len(loci), np.nan)
    got2 = np.full(len(loci), np.nan)
    order = []

    def begin(k):
        s = spans[k]
        blocks, anchors, stops = s["blocks"].copy(), s["anchors"].copy(), s["anchor_stop"].copy()
        ctx.span_stage_begin(s["comp"], blocks, anchors, stops, (k * 3 + 7) % 8)
        blocks[...] = 0  # the library has its own copy
        anchors[...] = 0
        stops[...] = 0

    begin(0)
    for k in range(len(spans)):
        if k + 1 < len(spans):
            begin(k + 1)  # the second one in flight
        slot = (k * 3 + 7) % 8
        ctx.span_stage_wait(slot)
        s = spans[k]
        rc, _stats = ctx.call_span_deferred(s["comp"], s["blocks"], s["anchors"], s["anchor_stop"], s["locus_tid"], s["locus_start"], s["locus_end"],
                                            minlen, support, unphased, stage_slot=slot, prestaged=True)
        assert rc == 0
        order.extend(int(i) for i in s["locus_index"])
    rc, p1, p2, _ties, _ms = ctx.call_flush()
    assert rc == 0 and len(p1) == len(order)
    got1[order], got2[order] = p1, p2
    want1, want2 = _expected(loci, recs, unphased, minlen, support)
    assert gen.same_f64(got1, want1) and gen.same_f64(got2, want2)


@pytest.mark.parametrize("seed,unphased,span_bytes,gap,flush_every", [(1, False, 3_000, None, 0), (2, True, 20_000, None, 3), (3, False, 3_000, None, 2),
                                                                      (6, True, 30_000, 0, 0), (7, False, 1, None, 5)])
def test_deferred_spans_equal_span_by_span_calls(ctx, tmp_path, monkeypatch, seed, unphased, span_bytes, gap, flush_every):
    """inq_call_span_deferred + inq_call_flush (the locus kernels once over the batches of several spans: what the CLI does so
    that a launch holds enough loci) against inq_call_span span by span: the same rows, locus for locus, and - fetched back -
    the concatenation of the spans' batches; staged and unstaged spans mixed; buffers growing while they hold earlier spans."""
    from inquistr_amd import call
    from tests import gen
    from tests.test_host_frontend import _expected, _make_case
    from tools import bamio

    if gap is not None:
        monkeypatch.setenv("INQ_SPAN_GAP_BYTES", str(gap))
    minlen, support = 5, [3, 1, 2, 3][seed % 4]
    bam, bed, loci, recs = _make_case(tmp_path, seed, ultra_long=(seed == 3), block=bamio.BLOCK if gap is None else 1500)
    sp = call.Spans(bam, region_file=bed, minlen=minlen, support=support, threads=2, unphased=unphased, max_comp_bytes=span_bytes)
    spans = list(sp.spans())
    sp.close()
    assert len(spans) >= 2
    args = lambda s: (s["comp"], s["blocks"], s["anchors"], s["anchor_stop"], s["locus_tid"], s["locus_start"], s["locus_end"], minlen, support, unphased)
    one1, one2 = np.full(len(loci), np.nan), np.full(len(loci), np.nan)
    per_span = []
    for s_ in spans:
        rc, p1, p2, ties, stats = ctx.call_span(*args(s_))
        one1[s_["locus_index"]], one2[s_["locus_index"]] = p1, p2
        per_span.append((ctx.span_fetch_batch(stats, len(s_["locus_index"])), len(s_["locus_index"])))
    got1, got2 = np.full(len(loci), np.nan), np.full(len(loci), np.nan)
    waiting, batch_of = [], []

    def flush():
        n = ctx.deferred_loci
        assert n == sum(len(spans[k]["locus_index"]) for k in waiting)
        rc, p1, p2, ties, ms = ctx.call_flush()
        assert rc == 0 and len(p1) == n and ctx.deferred_loci == 0
        idx = np.concatenate([spans[k]["locus_index"] for k in waiting]) if waiting else np.zeros(0, dtype=np.int64)
        got1[idx], got2[idx] = p1, p2
        # the accumulated batch = the spans' batches one behind the other (offsets rebased)
        class St:  # sizes of the accumulated batch for span_fetch_batch
            n_cigar_words = sum(len(per_span[k][0][0]) for k in waiting)
            n_reads = sum(len(per_span[k][0][1]) for k in waiting)
            n_pairs = sum(len(per_span[k][0][2]) for k in waiting)
        if n:
            dev = ctx.span_fetch_batch(St, n)
            j = 0
            for k in waiting:
                for jj in range(per_span[k][1]):
                    assert _locus_view(dev, j) == _locus_view(per_span[k][0], jj), (seed, k, jj)
                    j += 1
        waiting.clear()

    for k, s_ in enumerate(spans):
        rc, stats = ctx.call_span_deferred(*args(s_), stage_slot=(k % 3) if k % 2 else None)
        assert rc == 0
        waiting.append(k)
        if flush_every and (k + 1) % flush_every == 0:
            flush()
    flush()
    flush()  # nothing waits: an empty flush is fine
    assert gen.same_f64(got1, one1) and gen.same_f64(got2, one2)
    want1, want2 = _expected(loci, recs, unphased, minlen, support)
    assert gen.same_f64(got1, want1) and gen.same_f64(got2, want2)
    # parameters must not change inside a batch
    ctx.call_span_deferred(*args(spans[0]))
    bad = list(args(spans[1]))
    bad[7] = minlen + 1
    rc, _ = ctx.call_span_deferred(*bad, check=False)
    assert rc == hipcall.INQ_ERR_ARG
    ctx.call_flush()


@pytest.mark.parametrize("seed,unphased,threads", [(11, False, 1), (12, True, 4)])
def test_device_front_end_text_equals_host_front_end(tmp_path, seed, unphased, threads):
    from inquistr_amd import call
    from tests.test_host_frontend import _make_case

    bam, bed, loci, recs = _make_case(tmp_path, seed, n_loci=120)
    texts = {}
    for fe in ("host", "device"):
        path = tmp_path / f"{fe}.inq"
        with open(path, "w") as f:
            call.genotype_repeats(bam, None, bed, 5, 3, threads, unphased, None, None, out=f, frontend=fe)
        texts[fe] = path.read_text()
    assert texts["host"] == texts["device"] and texts["host"].count("\n") == len(loci) + 1


@pytest.mark.skipif(not __import__("tools.libdeflate_shim", fromlist=["x"]).available(), reason="libdeflate runtime not in the image")
@pytest.mark.parametrize("level,unphased", [(6, False), (12, True), (1, False)])
def test_bam_written_with_libdeflate_through_both_front_ends(tmp_path, monkeypatch, level, unphased):
    """A BAM whose BGZF blocks libdeflate compressed (what an htslib built with libdeflate writes; small blocks so that records
    straddle many of them): the device front end's text equals the host sweep's (zlib inflate) and the Python restatement's."""
    from inquistr_amd import call
    from tests.test_gpu_end_to_end import _expected_text
    from tests.test_host_frontend import _make_case
    from tools import bamio
    from tools import libdeflate_shim as ld

    comp = ld.Compressor(level)
    monkeypatch.setattr(bamio, "bgzf_block", lambda data, lv=1: ld.bgzf_block(data, level, comp))
    bam, bed, loci, recs = _make_case(tmp_path, 40 + level, n_loci=110, ultra_long=True, block=3000 if level == 12 else bamio.BLOCK)
    monkeypatch.undo()
    texts = {}
    for fe in ("host", "device"):
        path = tmp_path / f"{fe}.inq"
        with open(path, "w") as f:
            call.genotype_repeats(bam, None, bed, 5, 3, 4, unphased, "S", None, out=f, frontend=fe)
        texts[fe] = path.read_text()
    assert texts["host"] == texts["device"] == _expected_text(loci, recs, unphased, 5, 3, "S", 4)


def test_device_front_end_through_a_csi_index(tmp_path):
    """A BAM that only has a .csi next to it ([3P] IndexedReader::from_path takes either index, src/call.rs:242): spans planned from
    the .csi's bins and per-bin offsets; the text equals the host sweep's and the Python restatement's."""
    from inquistr_amd import call
    from tests.test_csi_index import _reindex
    from tests.test_gpu_end_to_end import _expected_text
    from tests.test_host_frontend import _make_case

    bam, bed, loci, recs = _make_case(tmp_path, 31, n_loci=80)
    csi_bam = _reindex(tmp_path, bam, recs, 14, 5, "only_csi.sorted.bam", block=3000)
    texts = {}
    for fe in ("host", "device"):
        path = tmp_path / f"{fe}.inq"
        with open(path, "w") as f:
            call.genotype_repeats(csi_bam, None, bed, 5, 3, 3, False, "S", None, out=f, frontend=fe)
        texts[fe] = path.read_text()
    assert texts["host"] == texts["device"] == _expected_text(loci, recs, False, 5, 3, "S", 3)


def test_device_front_end_error_classes(tmp_path):
    """The reference's panics that live in the record accessors keep their exit status through the device path."""
    from inquistr_amd import call
    from oracle import pyoracle as py
    from tools import bamio

    def run(recs, unphased=False, sort=True, tags=None):
        bam = str(tmp_path / "e.bam")
        w = bamio.BamWriter(bam, [("chr1", 100000)])
        for i, r in enumerate(sorted(recs, key=lambda r: r.pos) if sort else recs):
            w.add(f"r{i}", r.flag, 0, r.pos, r.mapq, r.cigar, (tags or (lambda r: [("HP", r.hp[0], r.hp[1])] + ([("SA", r.sa[0], r.sa[1])] if r.sa else [])))(r))
        w.close()
        with open(tmp_path / "e.inq", "w") as f:
            call.genotype_repeats(bam, "chr1:5000-5050", None, 5, 3, 1, unphased, None, None, out=f, frontend="device")
        return (tmp_path / "e.inq").read_text()

    ok = [py.Record(pos=4800, cigar=[("M", 210), ("I", 12), ("M", 300)], hp=("C", 1 + k % 2)) for k in range(8)]
    text = run(ok)
    assert text.splitlines()[1] == "chr1\t5000\t5050\t12\t12"
    # HP typed 's': get_phase panics, but only in phased mode
    bad_hp = ok + [py.Record(pos=4900, cigar=[("M", 400)], hp=("s", 1))]
    with pytest.raises(call.CallError) as e:
        run(bad_hp)
    assert e.value.status == 101
    assert run(bad_hp, unphased=True).count("\n") == 2
    # SA that is not a string on a read with a soft clip
    bad_sa = ok + [py.Record(pos=4990, cigar=[("S", 20), ("M", 400)], hp=("C", 1), sa=("i", 7))]
    with pytest.raises(call.CallError) as e:
        run(bad_sa)
    assert e.value.status == 101
    # ... and a malformed SA string
    bad_sa2 = ok + [py.Record(pos=4990, cigar=[("S", 20), ("M", 400)], hp=("C", 1), sa=("Z", "chr1,notanumber,-,50M,60,0;"))]
    with pytest.raises(call.CallError) as e:
        run(bad_sa2)
    assert e.value.status == 101
    # the same read filtered out (mapq <= 10) never reaches call_from_cigar, hence never is_accidental_2d
    dropped = ok + [py.Record(pos=4990, cigar=[("S", 20), ("M", 400)], mapq=5, hp=("C", 1), sa=("Z", "chr1,notanumber,-,50M,60,0;"))]
    assert run(dropped).splitlines()[1] == "chr1\t5000\t5050\t12\t12"
    # the same reads without a soft clip never reach is_accidental_2d
    fine = ok + [py.Record(pos=4990, cigar=[("M", 400)], hp=("C", 1), sa=("Z", "chr1,notanumber,-,50M,60,0;"))]
    assert run(fine).count("\n") == 2
    # records out of coordinate order
    with pytest.raises(call.CallError) as e:
        run([ok[0], py.Record(pos=4700, cigar=[("M", 500)], hp=("C", 1))] + ok[1:], sort=False)
    assert e.value.status == 101


@pytest.mark.parametrize("frontend", ["host", "device"])
def test_error_class_sweep(tmp_path, frontend):
    """{mapq 5/60} x {HP absent/C/i/s} x {spanning/inside/partial} x {clip/no clip} x {10 SA shapes} x {phased, unphased}:
    exit status AND row text against the Python restatement.  is_accidental_2d can only panic for a read that passed
    the filter (src/call.rs:303,357 -> :394), get_phase for any fetched read in phased mode (:349)."""
    from inquistr_amd import call
    from tests import errclass

    bam = str(tmp_path / "e.bam")
    out = tmp_path / "e.inq"
    region = "%s:%d-%d" % errclass.LOCUS
    n_panic = n_rows = 0
    for name, probe in errclass.cases():
        recs = errclass.write_bam(bam, errclass.good_reads() + [probe])
        for unphased in (False, True):
            want = errclass.expected(recs, unphased)
            try:
                with open(out, "w") as f:
                    call.genotype_repeats(bam, region, None, 5, 3, 1, unphased, None, None, out=f, frontend=frontend)
                got = out.read_text().splitlines()[1]
            except call.CallError as e:
                assert e.status == 101, (name, unphased, e)
                got = None
            assert got == want, (name, unphased, frontend)
            n_panic += want is None
            n_rows += want is not None
    assert n_panic == 4 * 6 + 2 * 3 * 2 * 10 + 2 * 2 * 6 and n_rows == 2 * 480 - n_panic


def test_device_front_end_region_string_and_long_cigar_tag(tmp_path):
    """-r with one locus; reads with 70 000 CIGAR ops (real CIGAR in CG:B,I behind <l_seq>S<ref>N) through the
    device record scan; both front ends must print the row the Python restatement gives."""
    from inquistr_amd import call
    from oracle import pyoracle as py
    from tests import gen
    from tools import bamio

    rng = random.Random(9)
    big = gen.random_cigar(rng, 70_000)
    span = py.reference_end(py.Record(pos=0, cigar=big))
    start = 1000 + span // 2
    recs = [py.Record(pos=1000, cigar=big, mapq=60, hp=("C", 1), tid=0) for _ in range(3)]
    recs += [py.Record(pos=start - 200, cigar=[("M", 150), ("I", 30 + k), ("M", 400)], hp=("i", 2), tid=0) for k in range(3)]
    recs.sort(key=lambda r: r.pos)
    bam = str(tmp_path / "long.bam")
    w = bamio.BamWriter(bam, [("chr7", span + 100_000)])
    for i, r in enumerate(recs):
        w.add(f"r{i}", 0, 0, r.pos, 60, r.cigar, [("HP", r.hp[0], r.hp[1])], l_seq=5)
    w.close()
    a, b = py.genotype_repeat_phased(recs, 0, start, start + 100, 5, 3)
    want = py.format_header("S") + "\n" + py.format_row("chr7", start, start + 100, a, b) + "\n"
    for fe in ("host", "device"):
        out = tmp_path / f"{fe}.inq"
        with open(out, "w") as f:
            call.genotype_repeats(bam, f"chr7:{start}-{start + 100}", None, 5, 3, 1, False, "S", None, out=f, frontend=fe)
        assert out.read_text() == want, fe


@pytest.mark.parametrize("workload,loci", [("phased10k", 3000), ("expansion50k", 1500)])
def test_device_front_end_on_synthetic_workloads(tmp_path, workload, loci):
    """The benchmark's BAM generator at a few thousand loci: HP-phased reads, soft clips, 2000-op reads;
    device front end = host front end, byte for byte, with -t 1 (BED order) and -t 4 (sorted)."""
    from inquistr_amd import call, synth
    from tools import make_synth_bam

    prefix = str(tmp_path / "w")
    make_synth_bam.write(workload, loci, prefix)
    wl = synth.WORKLOADS[workload]
    for threads in (1, 4):
        texts = {}
        for fe in ("host", "device"):
            out = tmp_path / f"{fe}{threads}.inq"
            with open(out, "w") as f:
                call.genotype_repeats(prefix + ".bam", None, prefix + ".bed", wl.minlen, wl.support, threads, wl.unphased, "S", None,
                                      out=f, frontend=fe)
            texts[fe] = out.read_text()
        assert texts["host"] == texts["device"] and texts["host"].count("\n") == loci + 1
        assert "NaN\tNaN" not in texts["host"].split("\n", 2)[1]


def test_call_span_rejects_malformed_spans(ctx, tmp_path):
    """Shape errors are caught on the host before any kernel runs; device-detected ones come back as codes."""
    from inquistr_amd import call
    from tests.test_host_frontend import _make_case

    bam, bed, loci, recs = _make_case(tmp_path, 21, n_loci=20)
    span = next(iter(call.Spans(bam, region_file=bed).spans()))
    args = [span[k] for k in ("comp", "blocks", "anchors", "anchor_stop", "locus_tid", "locus_start", "locus_end")]

    def run(**kw):
        a = dict(zip(("comp", "blocks", "anchors", "anchor_stop", "locus_tid", "locus_start", "locus_end"), args))
        a.update(kw)
        return ctx.call_span(a["comp"], a["blocks"], a["anchors"], a["anchor_stop"], a["locus_tid"], a["locus_start"], a["locus_end"],
                             5, 3, False, check=False)[0]

    assert run() == 0
    # anchors out of order / behind the inflated bytes / stop in front of its anchor
    bad = span["anchors"].copy()
    bad[[0, 1]] = bad[[1, 0]]
    assert run(anchors=bad) == hipcall.INQ_ERR_ARG
    stops = span["anchor_stop"].copy()
    stops[0] = 0
    assert run(anchor_stop=stops) == hipcall.INQ_ERR_ARG
    # block table that is not dense, or points outside comp
    blocks = span["blocks"].copy()
    blocks["out_off"][1] += 1
    assert run(blocks=blocks) == hipcall.INQ_ERR_ARG
    blocks = span["blocks"].copy()
    blocks["comp_len"][-1] += 1 << 20
    assert run(blocks=blocks) == hipcall.INQ_ERR_ARG
    # locus out of domain, negative contig
    ls = span["locus_start"].copy()
    ls[0] = 5
    assert run(locus_start=ls) == hipcall.INQ_ERR_LOCUS
    lt = span["locus_tid"].copy()
    lt[0] = -1
    assert run(locus_tid=lt) == hipcall.INQ_ERR_ARG
    # an anchor that is not a record start: the chain does not land on the next anchor
    bad = span["anchors"].copy()
    if len(bad) > 2:
        bad[1] += 1
        assert run(anchors=bad) == hipcall.INQ_ERR_BAM
    # a flipped payload byte: CRC32 (or the inflate itself) notices
    comp = span["comp"].copy()
    comp[int(span["blocks"]["comp_off"][0]) + 30] ^= 0x10
    assert run(comp=comp) == hipcall.INQ_ERR_INFLATE
    assert run() == 0  # the ctx is usable after every error
    # a staged slot only serves the span that was staged into it
    a = dict(zip(("comp", "blocks", "anchors", "anchor_stop", "locus_tid", "locus_start", "locus_end"), args))
    assert ctx.call_span(*[a[k] for k in ("comp", "blocks", "anchors", "anchor_stop", "locus_tid", "locus_start", "locus_end")],
                         5, 3, False, check=False, stage_slot=1)[0] == 0
    other = hipcall.SpanC()
    assert ctx._L.inq_call_span_staged(ctx._h, other, 2, None, None) == hipcall.INQ_ERR_ARG


def test_inflate_fuzz_agrees_with_zlib_on_mutated_streams(ctx):
    """Thousands of damaged DEFLATE payloads: the kernel must accept exactly what zlib accepts (stream ends, ISIZE
    bytes produced), with the same bytes, and flag everything else - without faulting or hanging on any of them."""
    rng = random.Random(2024)
    base = []
    for k in range(24):
        n = rng.choice([40, 300, 2000, 9000])
        kind = k % 4
        if kind == 0:
            data = bytes(rng.choice(b"ACGTN=") for _ in range(n))
        elif kind == 1:
            data = bytes(rng.getrandbits(8) for _ in range(n))
        elif kind == 2:
            data = (bytes(rng.getrandbits(8) for _ in range(17)) * (n // 17 + 1))[:n]
        else:
            data = b"".join(struct.pack("<I", rng.randint(1, 400) << 4 | rng.choice([0, 1, 2, 4])) for _ in range(n // 4))
        level, strategy = rng.choice([(1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                                      (6, zlib.Z_FIXED), (0, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_HUFFMAN_ONLY)])
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        base.append((co.compress(data) + co.flush(), len(data)))
    payloads, isizes = [], []
    for _ in range(4000):
        p, n = rng.choice(base)
        p = bytearray(p)
        for _ in range(rng.choice([0, 1, 1, 1, 2, 5])):
            how = rng.random()
            at = rng.randrange(len(p))
            if how < 0.6:
                p[at] ^= 1 << rng.randrange(8)
            elif how < 0.8:
                p[at] = rng.getrandbits(8)
            elif how < 0.9 and len(p) > 8:
                del p[rng.randrange(len(p) // 2, len(p)):]  # truncate
            else:
                p[at:at] = bytes(rng.getrandbits(8) for _ in range(rng.randint(1, 4)))  # insert
        payloads.append(bytes(p))
        isizes.append(n if rng.random() < 0.9 else max(0, n + rng.choice([-1, 1, 7])))
    # lay the payloads out like BGZF blocks: payload + 8 trailer bytes (CRC not checked here)
    comp = bytearray()
    blocks = np.zeros(len(payloads), dtype=hipcall.BGZF_BLOCK_DTYPE)
    uo = 0
    for i, (p, n) in enumerate(zip(payloads, isizes)):
        blocks[i] = (len(comp), len(p), n, uo)
        comp += p + bytes(8)
        uo += n
    ctx.set_option("verify_crc", 0)
    try:
        rc, out, status = ctx.bgzf_inflate(bytes(comp), blocks, check=False)
    finally:
        ctx.set_option("verify_crc", 1)
    n_ok = 0
    for i, (p, n) in enumerate(zip(payloads, isizes)):
        d = zlib.decompressobj(-15)
        try:
            got = d.decompress(p, n + 1)  # one byte more than the block may hold
            accept = d.eof and len(got) == n
        except zlib.error:
            accept, got = False, b""
        o = int(blocks[i]["out_off"])
        if accept:
            n_ok += 1
            assert status[i] == 0, (i, hex(int(status[i])))
            assert out[o : o + n].tobytes() == got, i
        else:
            assert status[i] != 0, (i, len(p), n)
    assert 400 < n_ok < 3600  # the corpus exercises both outcomes
    assert (rc == 0) == (n_ok == len(payloads))


def _bam_like_payloads(rng: random.Random):
    """What the blocks of a long-read BAM hold: fixed fields + CIGARs, packed bases, base qualities, ML / MM tags."""
    out = []
    nib = [1, 2, 4, 8]
    for kind in range(4):
        parts = []
        while sum(len(x) for x in parts) < 65000:
            if kind in (0, 3):  # records without SEQ: fixed fields, read name, CIGAR words
                parts.append(struct.pack("<iiBBHHHIiii", 0, rng.randint(1, 1 << 27), 9, 60, 4681, rng.randint(50, 300), 16 * rng.randint(0, 1), 0, -1, -1, 0))
                parts.append(b"read%06d\0" % rng.randint(0, 999999))
                parts.append(b"".join(struct.pack("<I", rng.randint(1, 400) << 4 | rng.choice([0, 0, 0, 1, 2, 7, 8])) for _ in range(rng.randint(50, 300))))
                if kind == 3:
                    parts.append(b"HPC" + bytes([rng.randint(1, 2)]) + b"SAZchr7,%d,+,50M,60,0;\0" % rng.randint(1, 1 << 27))
            elif kind == 1:  # packed bases
                parts.append(bytes(rng.choice(nib) << 4 | rng.choice(nib) for _ in range(6000)))
            else:  # qualities (skewed Phred) and a methylation tag
                parts.append(bytes(min(50, max(1, int(rng.gammavariate(4.0, 5.0)))) for _ in range(6000)))
                parts.append(b"MLBC" + struct.pack("<I", 300) + bytes(rng.choice([0, 3, 250, 255]) for _ in range(300)))
        out.append(b"".join(parts)[:65280])
    return out


@pytest.mark.skipif(not __import__("tools.libdeflate_shim", fromlist=["x"]).available(), reason="libdeflate runtime not in the image")
def test_inflate_of_libdeflate_streams(ctx):
    """htslib is commonly built with libdeflate: BAMs written by it hold DEFLATE streams zlib's compressor never produces (other block
    splitting, lazy / near-optimal match choices from level 8 on, other code shapes).  Every payload class of the zlib test and
    BAM-like bytes, compressed by the image's libdeflate at levels 0 - 12, through every form of the device inflate (the fixture's):
    the bytes of the input, and what libdeflate's own decoder and zlib's make of the same stream."""
    from tools import libdeflate_shim as ld

    rng = random.Random(77)
    blobs, want = [], []
    for data in _payloads(rng) + _bam_like_payloads(rng):
        for level in (0, 1, 2, 3, 5, 6, 7, 8, 9, 10, 12):
            if level == 0 and len(data) > 65000:
                data = data[:65000]
            try:
                blob = ld.bgzf_block(data, level)
            except AssertionError:
                continue  # incompressible bytes that do not fit one block at this level
            payload = blob[18:-8]
            assert zlib.decompressobj(-15).decompress(payload) == data
            rc, back = ld.decompress(payload, len(data))
            assert rc == 0 and back == data
            blobs.append(blob)
            want.append(data)
    comp = b"".join(blobs)
    blocks = hipcall.scan_bgzf(comp)
    assert len(blocks) == len(want) and len(want) > 100
    rc, out, status = ctx.bgzf_inflate(comp, blocks)
    assert rc == 0 and not status.any()
    for b, w in zip(blocks, want):
        got = out[int(b["out_off"]) : int(b["out_off"]) + int(b["isize"])].tobytes()
        assert got == w, (len(w), w[:16])


@pytest.mark.skipif(not __import__("tools.libdeflate_shim", fromlist=["x"]).available(), reason="libdeflate runtime not in the image")
def test_inflate_fuzz_on_mutated_libdeflate_streams(ctx):
    """The mutation fuzz with libdeflate-written originals (levels 1 - 12): accept exactly what zlib accepts, with the same bytes."""
    from tools import libdeflate_shim as ld

    rng = random.Random(4711)
    base = []
    for k in range(24):
        n = rng.choice([40, 300, 2000, 9000, 30000])
        kind = k % 4
        if kind == 0:
            data = bytes(rng.choice(b"ACGTN=") for _ in range(n))
        elif kind == 1:
            data = bytes(min(50, max(1, int(rng.gammavariate(4.0, 5.0)))) for _ in range(n))
        elif kind == 2:
            data = (bytes(rng.getrandbits(8) for _ in range(17)) * (n // 17 + 1))[:n]
        else:
            data = b"".join(struct.pack("<I", rng.randint(1, 400) << 4 | rng.choice([0, 1, 2, 4])) for _ in range(n // 4))
        base.append((ld.Compressor(rng.choice([1, 3, 6, 6, 9, 12])).compress(data), len(data)))
    payloads, isizes = [], []
    for _ in range(3000):
        p, n = rng.choice(base)
        p = bytearray(p)
        for _ in range(rng.choice([0, 1, 1, 1, 2, 5])):
            how = rng.random()
            at = rng.randrange(len(p))
            if how < 0.6:
                p[at] ^= 1 << rng.randrange(8)
            elif how < 0.8:
                p[at] = rng.getrandbits(8)
            elif how < 0.9 and len(p) > 8:
                del p[rng.randrange(len(p) // 2, len(p)):]
            else:
                p[at:at] = bytes(rng.getrandbits(8) for _ in range(rng.randint(1, 4)))
        payloads.append(bytes(p))
        isizes.append(n if rng.random() < 0.9 else max(0, n + rng.choice([-1, 1, 7])))
    comp = bytearray()
    blocks = np.zeros(len(payloads), dtype=hipcall.BGZF_BLOCK_DTYPE)
    uo = 0
    for i, (p, n) in enumerate(zip(payloads, isizes)):
        blocks[i] = (len(comp), len(p), n, uo)
        comp += p + bytes(8)
        uo += n
    ctx.set_option("verify_crc", 0)
    try:
        rc, out, status = ctx.bgzf_inflate(bytes(comp), blocks, check=False)
    finally:
        ctx.set_option("verify_crc", 1)
    n_ok = 0
    for i, (p, n) in enumerate(zip(payloads, isizes)):
        d = zlib.decompressobj(-15)
        try:
            got = d.decompress(p, n + 1)
            accept = d.eof and len(got) == n
        except zlib.error:
            accept, got = False, b""
        o = int(blocks[i]["out_off"])
        if accept:
            n_ok += 1
            assert status[i] == 0, (i, hex(int(status[i])))
            assert out[o : o + n].tobytes() == got, i
        else:
            assert status[i] != 0, (i, len(p), n)
    assert 300 < n_ok < 2800


def test_device_aux_walk_skips_long_strings_of_every_length(tmp_path):
    """HP behind a Z tag of every length 0..70 and of 20 000 bytes (methylation strings sit in front of the HP tag
    phasing tools append): the word-at-a-time NUL search must land on the next tag for every alignment."""
    from inquistr_amd import call
    from tools import bamio

    lens = list(range(0, 71)) + [20_000, 20_001, 20_002, 20_003]
    bam = str(tmp_path / "aux.bam")
    w = bamio.BamWriter(bam, [("chr1", 100000)])
    for i, n in enumerate(lens):
        hp = 1 + i % 2
        w.add(f"r{i}", 0, 0, 4800, 60, [("M", 210), ("I", 12 if hp == 1 else 30), ("M", 300)],
              [("ML", "B", ("C", [7] * (i % 5))), ("MM", "Z", "C+m," * (n // 4) + "x" * (n % 4)), ("HP", "C", hp), ("SA", "Z", "chr1,1,+,5M,0,0;")],
              l_seq=70_000 if i % 10 == 0 else i % 3)  # some records longer than a whole BGZF block
    w.close()
    texts = {}
    for fe in ("host", "device"):
        out = tmp_path / f"{fe}.inq"
        with open(out, "w") as f:
            call.genotype_repeats(bam, "chr1:5000-5050", None, 5, 3, 1, False, "S", None, out=f, frontend=fe)
        texts[fe] = out.read_text()
    assert texts["device"] == texts["host"] == "chromosome\tbegin\tend\tS_H1\tS_H2\nchr1\t5000\t5050\t12\t30\n"


def test_edge_loci_through_both_front_ends(tmp_path):
    """Loci at the lowest legal start (10), at the end of a contig, duplicated, zero-length, nested and abutting;
    reads starting at position 0 and ending on the last base: device front end = host front end = the Python
    restatement over every record."""
    from inquistr_amd import call
    from oracle import pyoracle as py
    from tests import gen
    from tools import bamio

    rng = random.Random(31)
    LN = 60_000
    loci = [("chrA", 10, 40), ("chrA", 10, 40), ("chrA", 25, 25), ("chrA", 20, 300), ("chrA", 300, 320), ("chrA", 320, 340),
            ("chrA", LN - 200, LN - 1), ("chrA", LN - 11, LN - 1), ("chrB", 10, 10), ("chrB", 5000, 5050)]
    recs = {0: [], 1: []}
    for t, (lo, hi) in ((0, (0, 400)), (0, (LN - 900, LN - 1)), (1, (0, 60)), (1, (4700, 5300))):
        for k in range(40):
            pos = rng.randint(lo, max(lo, hi - 50)) if k % 5 else lo  # some reads start on the first base of the range
            cig = gen.random_cigar(rng, rng.choice([1, 3, 9, 30]))
            ref = sum(n for o, n in cig if o in "MDN=X")
            room = (LN if t == 0 else 20_000) - pos
            if ref > room:
                cig = [("M", max(room, 1))]
            recs[t].append(py.Record(pos=pos, cigar=cig, mapq=rng.choice([5, 20, 60]), flag=rng.choice([0, 16]), hp=("C", rng.choice([1, 2])), tid=t))
    recs[0].append(py.Record(pos=LN - 700, cigar=[("M", 350), ("I", 9), ("M", 350)], mapq=60, hp=("C", 1), tid=0))  # ends on the last base
    bam = str(tmp_path / "edge.bam")
    w = bamio.BamWriter(bam, [("chrA", LN), ("chrB", 20_000)], block=3000)
    k = 0
    for t in (0, 1):
        recs[t].sort(key=lambda r: r.pos)
        for r in recs[t]:
            w.add(f"r{k}", r.flag, t, r.pos, r.mapq, r.cigar, [("HP", r.hp[0], r.hp[1])], l_seq=k % 4)
            k += 1
    w.close()
    bed = tmp_path / "edge.bed"
    bed.write_text("".join(f"{c}\t{s}\t{e}\n" for c, s, e in loci))
    for unphased in (False, True):
        rows = [py.format_header("S")]
        for c, s, e in loci:
            t = 0 if c == "chrA" else 1
            if unphased:
                a, b, _ = py.genotype_repeat_unphased(recs[t], t, s, e, 5, 2)
            else:
                a, b = py.genotype_repeat_phased(recs[t], t, s, e, 5, 2)
            rows.append(py.format_row(c, s, e, a, b))
        want = "\n".join(rows) + "\n"
        for fe in ("host", "device"):
            out = tmp_path / f"{fe}.inq"
            with open(out, "w") as f:
                call.genotype_repeats(bam, None, str(bed), 5, 2, 1, unphased, "S", None, out=f, frontend=fe)
            assert out.read_text() == want, (fe, unphased)


def test_many_spans_with_staged_uploads_equal_the_host_front_end(tmp_path, monkeypatch):
    """20 000 loci / 200 MB of BAM cut into 64 MB spans: the loader thread stages span k+1 on the device while span k
    is being inflated.  Same bytes as the host front end, every row called, and the same again with one big span."""
    from inquistr_amd import call
    from tools import make_synth_bam

    prefix = str(tmp_path / "w")
    make_synth_bam.write("unphased100k", 20_000, prefix)
    texts = {}
    L = call.load()
    try:
        # (the span is inflated when it is staged - option "inflate_ahead", the default - or when it is called: set for the contexts
        # the host library makes through inq_host_ctx_option, the library reads no such switch from the environment)
        for name, env, ahead in (("host", {"INQ_FRONTEND": "host"}, 1), ("device64", {"INQ_FRONTEND": "device", "INQ_SPAN_MB": "64"}, 0),
                                 ("device64_inflated_when_staged", {"INQ_FRONTEND": "device", "INQ_SPAN_MB": "64"}, 1),
                                 ("device", {"INQ_FRONTEND": "device"}, 1), ("auto", {}, 1)):
            for k in ("INQ_FRONTEND", "INQ_SPAN_MB"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            assert L.inq_host_ctx_option(b"inflate_ahead", ahead) == 0
            out = tmp_path / f"{name}.inq"
            with open(out, "w") as f:
                call.genotype_repeats(prefix + ".bam", None, prefix + ".bed", 5, 3, 8, True, "S", None, out=f)
            texts[name] = out.read_text()
    finally:
        L.inq_host_ctx_option(b"inflate_ahead", 1)
    assert L.inq_host_ctx_option(b"no_such_option", 1) == 1 and L.inq_host_ctx_option(b"inflate_algo", 7) == 1
    assert texts["host"] == texts["device64"] == texts["device64_inflated_when_staged"] == texts["device"] == texts["auto"]
    rows = texts["host"].splitlines()
    assert len(rows) == 20_001 and not any(r.endswith("NaN\tNaN") for r in rows[1:])
