"""The host-side probe that picks the form of the workgroup inflate's symbol loop (inquistr_amd/csrc/deflate_probe.h): the
literals' share of a deflate block's code space, read from its dynamic-Huffman header (RFC 1951 3.2.7), against a plain-Python
reading of the same header; and the decision on the four kinds of data of tools/inflate_bench.py.  No GPU involved."""
import ctypes as C
import os
import random
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("probe") / "libprobe.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-o", so, os.path.join(ROOT, "tools", "deflate_probe_shim.cc")])
    L = C.CDLL(so)
    L.inq_probe_literal_mass.restype = C.c_int
    L.inq_probe_literal_mass.argtypes = [C.c_char_p, C.c_size_t]
    L.inq_probe_wants_pairs.restype = C.c_uint32
    L.inq_probe_wants_pairs.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64]
    return L


class _Bits:
    def __init__(self, d):
        self.d, self.p = d, 0

    def take(self, n):
        v = 0
        for i in range(n):
            v |= ((self.d[self.p >> 3] >> (self.p & 7)) & 1) << i
            self.p += 1
        return v


def _py_mass(payload):
    """RFC 1951 3.2.7 read the slow way: literal/length code lengths of the first block, or None if it is not dynamic."""
    br = _Bits(payload)
    br.take(1)
    if br.take(2) != 2:
        return None
    hlit, hdist, hclen = br.take(5) + 257, br.take(5) + 1, br.take(4) + 4
    order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
    cl = [0] * 19
    for i in range(hclen):
        cl[order[i]] = br.take(3)
    count = [0] * 8
    for l in cl:
        count[l] += 1
    count[0] = 0
    code, nxt = 0, [0] * 8
    for b in range(1, 8):
        code = (code + count[b - 1]) << 1
        nxt[b] = code
    table = {}
    for s, l in enumerate(cl):
        if l:
            table[(l, nxt[l])] = s
            nxt[l] += 1
    lens = []
    while len(lens) < hlit + hdist:
        c = l = 0
        while True:
            c = (c << 1) | br.take(1)
            l += 1
            if (l, c) in table:
                s = table[(l, c)]
                break
        if s < 16:
            lens.append(s)
        elif s == 16:
            lens += [lens[-1]] * (3 + br.take(2))
        elif s == 17:
            lens += [0] * (3 + br.take(3))
        else:
            lens += [0] * (11 + br.take(7))
    return sum(1 << (15 - l) for l in lens[:256] if l)


def test_mass_equals_a_plain_reading_of_the_header(probe):
    rng = random.Random(7)
    seen = set()
    for trial in range(60):
        kind = trial % 5
        n = rng.randrange(200, 60000)
        if kind == 0:
            data = bytes(rng.randrange(0, 51) for _ in range(n))
        elif kind == 1:
            data = b"".join(struct.pack("<I", rng.randint(1, 400) << 4 | rng.choice([0, 0, 1, 2])) for _ in range(n // 4))
        elif kind == 2:
            data = bytes(rng.choice(b"ACGTN") for _ in range(n))
        elif kind == 3:
            data = bytes(rng.getrandbits(8) for _ in range(n))
        else:
            data = (b"period" * 40 + bytes(rng.getrandbits(8) for _ in range(50))) * (n // 300 + 1)
        co = zlib.compressobj(rng.choice([1, 6, 9]), zlib.DEFLATED, -15, 8, rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY]))
        payload = co.compress(data) + co.flush()
        want = _py_mass(payload)
        got = probe.inq_probe_literal_mass(payload, len(payload))
        assert got == (-1 if want is None else want), (trial, kind)
        seen.add(want is None)
    assert seen == {False, True}  # dynamic blocks and others (stored: incompressible bytes) both occurred
    # stored / fixed blocks, truncated and empty input: "unknown", never a crash
    for payload in (zlib.compressobj(0, zlib.DEFLATED, -15).compress(b"abc") + b"", zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_FIXED).compress(b"abcabc"), b"", b"\x05"):
        assert probe.inq_probe_literal_mass(payload, len(payload)) == -1
    good = zlib.compressobj(6, zlib.DEFLATED, -15)
    payload = good.compress(bytes(rng.randrange(0, 51) for _ in range(5000))) + good.flush()
    full = probe.inq_probe_literal_mass(payload, len(payload))
    results = {probe.inq_probe_literal_mass(payload[:cut], cut) for cut in range(0, 80)}
    assert results == {-1, full}  # cut inside the header: unknown; behind the 256 literal lengths: the value (nothing else is read)


@pytest.mark.parametrize("kind,level,want", [("cigar", 1, 0), ("cigar", 6, 0), ("ont", 1, 1), ("ont", 6, 1), ("qual", 1, 1), ("seq", 1, 1), ("seq", 6, 1)])
def test_decision_on_the_benchmark_kinds(probe, kind, level, want):
    """CIGAR-only records are match-heavy (no pairs); sequence, quality and nanopore-like bytes are literal-heavy (pairs)."""
    from inquistr_amd import hipcall
    from tools import inflate_bench

    comp = inflate_bench.make_blocks(12, level, kind)
    blocks = hipcall.scan_bgzf(comp)
    assert len(blocks) == 12
    assert probe.inq_probe_wants_pairs(comp, len(comp), blocks.ctypes.data, len(blocks)) == want
    # nothing to look at: the default form
    assert probe.inq_probe_wants_pairs(comp, len(comp), blocks.ctypes.data, 0) == 1


def test_probe_reads_libdeflate_headers_too(probe):
    """htslib built with libdeflate writes other headers (other code-length code shapes, block splitting): the probe's reading equals
    the plain-Python one on them as well, every stream inflates to its input under zlib and under libdeflate's own decoder, and the
    decision on base-quality-like bytes is the one taken for zlib's streams (literal-heavy: the pairs form)."""
    from tools import libdeflate_shim as ld

    if not ld.available():
        pytest.skip("libdeflate runtime not in the image")
    rng = random.Random(11)
    seen = set()
    for trial in range(60):
        kind = trial % 4
        n = rng.randrange(200, 60000)
        if kind == 0:
            data = bytes(rng.randrange(0, 51) for _ in range(n))
        elif kind == 1:
            data = b"".join(struct.pack("<I", rng.randint(1, 400) << 4 | rng.choice([0, 0, 1, 2])) for _ in range(n // 4))
        elif kind == 2:
            data = bytes(rng.choice(b"ACGTN") for _ in range(n))
        else:
            data = bytes(rng.getrandbits(8) for _ in range(n))
        payload = ld.Compressor(rng.choice([1, 3, 6, 9, 12])).compress(data)
        assert zlib.decompressobj(-15).decompress(payload) == data
        rc, back = ld.decompress(payload, len(data))
        assert rc == 0 and back == data
        want = _py_mass(payload)
        got = probe.inq_probe_literal_mass(payload, len(payload))
        assert got == (-1 if want is None else want), (trial, kind)
        seen.add(want is None)
        if want is not None and kind == 0 and n > 4000:
            assert want / 32768.0 > 0.42, (trial, want)  # base qualities: literal-heavy, the pairs form
    assert False in seen
