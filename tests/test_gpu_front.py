"""Device front end (inq_bgzf_inflate / inq_call_span) against zlib, the host front end and the oracle."""
import os
import random
import struct
import zlib

import numpy as np
import pytest

from inquistr_amd import hipcall

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = hipcall.Context(0)
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
    if zlib_ok:
        assert status[1] == 0 and out[n : 2 * n].tobytes() == alt
    else:
        assert status[1] != 0 and rc == hipcall.INQ_ERR_INFLATE
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
