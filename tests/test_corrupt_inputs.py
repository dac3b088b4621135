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
