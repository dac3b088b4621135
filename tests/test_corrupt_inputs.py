"""Damaged inputs must end in an error status (the reference panics or exits on them), never in a crash: random byte flips and
truncations of the .bai, the .csi and the BAM itself through the host front end and the span planner (no GPU).  The same file
runs under ASan / UBSan in tests/test_sanitizers.py."""
import os
import random

import pytest

from inquistr_amd import call
from tests.test_csi_index import _reindex
from tests.test_host_frontend import _make_case


def _drain(bam, bed):
    """Opens and walks both host-side paths; returns "ok" or the exit status."""
    try:
        fe = call.FrontEnd(bam, region_file=bed, threads=2, max_batch_words=4000)
        for _batch, _idx in fe.batches():
            pass
        fe.close()
        sp = call.Spans(bam, region_file=bed, threads=2, max_comp_bytes=50_000)
        for _span in sp.spans():
            pass
        sp.close()
    except call.CallError as e:
        assert e.status in (1, 101), e
        return e.status
    return "ok"


@pytest.mark.parametrize("what", ["bai", "csi", "bam"])
def test_damaged_files_never_crash(tmp_path, what):
    rng = random.Random({"bai": 1, "csi": 2, "bam": 3}[what])
    bam, bed, loci, recs = _make_case(tmp_path, 51, n_loci=25)
    if what == "csi":
        bam = _reindex(tmp_path, bam, recs, 14, 5, "fuzz.sorted.bam")
    target = {"bai": bam + ".bai", "csi": bam + ".csi", "bam": bam}[what]
    good = open(target, "rb").read()
    assert _drain(bam, bed) == "ok"
    outcomes = set()
    for trial in range(int(os.environ.get("INQ_FUZZ_TRIALS", "24"))):
        data = bytearray(good)
        kind = trial % 4
        if kind == 0:  # a few flipped bytes
            for _ in range(rng.randint(1, 6)):
                data[rng.randrange(len(data))] ^= 1 << rng.randrange(8)
        elif kind == 1:  # truncated
            data = data[: rng.randrange(0, len(data))]
        elif kind == 2:  # a stretch overwritten with 0xff (huge counts and offsets)
            at = rng.randrange(len(data))
            data[at : at + rng.randint(1, 16)] = b"\xff" * min(16, len(data) - at)
        else:  # a stretch of zeros
            at = rng.randrange(len(data))
            data[at : at + rng.randint(1, 64)] = bytes(min(64, len(data) - at))
        open(target, "wb").write(bytes(data))
        outcomes.add(_drain(bam, bed))
    open(target, "wb").write(good)
    assert _drain(bam, bed) == "ok"
    assert outcomes - {"ok"}, "none of the damaged files was noticed"


@pytest.mark.parametrize("what", ["bai", "csi"])
@pytest.mark.parametrize("count", [0x7FFFFFFF, 200_000_000, 100_000_000])
@pytest.mark.parametrize("field", ["n_ref", "n_bin", "n_chunk"])
@pytest.mark.timeout(60)
def test_counts_the_file_cannot_hold_are_refused_at_once(tmp_path, what, count, field):
    """One 4-byte count of a valid index set to 10^8 .. 2^31 - 1 (n_ref = 2 * 10^8 built one BaiRef per claimed
    contig for 192 s and tens of GB before noticing; 2^31 - 1 ended in bad_alloc = exit 1).  htslib fails at once; the
    reference then panics in IndexedReader::from_path (src/call.rs:242-243): status 101, within the test's time limit."""
    import gzip
    import struct
    import time

    bam, bed, loci, recs = _make_case(tmp_path, 52, n_loci=5)
    if what == "csi":
        bam = _reindex(tmp_path, bam, recs, 14, 5, "counts.sorted.bam")
        raw = bytearray(gzip.open(bam + ".csi", "rb").read())
        l_aux = struct.unpack_from("<I", raw, 12)[0]
        at_ref = 16 + l_aux
        at_bin = at_ref + 4
        at_chunk = at_bin + 4 + 12  # first bin: bin u32, loffset u64, n_chunk u32
    else:
        raw = bytearray(open(bam + ".bai", "rb").read())
        at_ref, at_bin = 4, 8
        at_chunk = at_bin + 4 + 4  # first bin: bin u32, n_chunk u32
    at = {"n_ref": at_ref, "n_bin": at_bin, "n_chunk": at_chunk}[field]
    struct.pack_into("<I", raw, at, count)
    if what == "csi":
        from tools import bamio

        open(bam + ".csi", "wb").write(bamio.bgzf_block(bytes(raw)) + bamio.EOF_BLOCK)
    else:
        open(bam + ".bai", "wb").write(bytes(raw))
    t0 = time.perf_counter()
    with pytest.raises(call.CallError) as e:
        call.FrontEnd(bam, region_file=bed)
    assert e.value.status == 101, e.value
    assert time.perf_counter() - t0 < 5.0
