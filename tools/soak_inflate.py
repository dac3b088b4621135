#!/usr/bin/env python3
"""GPU box: the inflate fuzz of tests/test_gpu_front.py with other seeds (accept / reject and bytes vs zlib).
usage: python tools/soak_inflate.py [rounds of 4000 damaged streams] [rounds of 1500 valid blocks per kernel]"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_gpu_front as t  # noqa: E402
from inquistr_amd import hipcall  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
valid_rounds = int(sys.argv[2]) if len(sys.argv) > 2 else max(1, rounds // 2)  # 1500 blocks each, per kernel (slow to generate)
ctx = hipcall.Context(0)
orig = random.Random
for r in range(rounds):
    seed = int(os.environ.get("INQ_SOAK_SEED", "9000")) + r

    class Seeded(orig):  # the test builds its own Random(2024): give it another stream each round
        def __init__(self, _ignored=None):
            super().__init__(seed)

    random.Random = Seeded
    try:
        t.test_inflate_fuzz_agrees_with_zlib_on_mutated_streams(ctx)
    finally:
        random.Random = orig
    print(f"round {r}: 4000 damaged streams agree with zlib", flush=True)
    from tools import libdeflate_shim

    if libdeflate_shim.available():  # the same with originals written by libdeflate (levels 1 - 12)
        random.Random = Seeded
        try:
            t.test_inflate_fuzz_on_mutated_libdeflate_streams(ctx)
        finally:
            random.Random = orig
        print(f"round {r}: 3000 damaged libdeflate-written streams agree with zlib", flush=True)
print(f"inflate soak done: {rounds} rounds, 0 disagreements")

# ---- valid streams of many shapes, thousands of blocks per launch, both kernels, bytes against the input
import struct
import zlib

import numpy as np


def _blocks(rng, n):
    out = []
    for _ in range(n):
        kind = rng.randrange(9)
        size = rng.choice([0, 1, 2, 3, 17, 300, 5000, 30000, 65280, 65536]) if rng.random() < 0.3 else rng.randrange(1, 65281)
        if kind == 0:
            data = bytes(size)  # one symbol: the whole block out of one lane's segment
        elif kind == 1:
            data = bytes(rng.getrandbits(8) for _ in range(min(size, 20000)))  # incompressible
        elif kind == 2:
            period = bytes(rng.getrandbits(8) for _ in range(rng.choice([1, 2, 3, 4, 5, 7, 8, 13, 64, 300])))
            data = (period * (size // len(period) + 1))[:size]
        elif kind == 3:
            words = [bytes(rng.getrandbits(8) for _ in range(rng.randint(1, 12))) for _ in range(rng.randint(2, 400))]
            data = b"".join(rng.choice(words) for _ in range(size // 6 + 1))[:size]
        elif kind == 4:  # CIGAR-like little-endian words
            data = b"".join(struct.pack("<I", rng.randint(1, 400) << 4 | rng.choice([0, 0, 1, 2])) for _ in range(size // 4 + 1))[:size]
        elif kind == 5:  # quality-like
            data = bytes(rng.randrange(0, 51) for _ in range(min(size, 30000)))
        elif kind == 6:  # long runs separated by noise: matches of every length
            data = b"".join(bytes([rng.getrandbits(8)]) * rng.randint(1, 600) + bytes(rng.getrandbits(8) for _ in range(rng.randint(0, 9)))
                            for _ in range(size // 200 + 1))[:size]
        elif kind == 7:  # far matches
            head = bytes(rng.getrandbits(8) for _ in range(rng.randint(3, 300)))
            data = (head + bytes(rng.getrandbits(8) for _ in range(rng.randint(0, 33000))) + head * rng.randint(1, 4))[:65280]
        else:
            data = bytes(rng.choice(b"ACGTN") for _ in range(min(size, 40000)))
        level = rng.choice([1, 1, 6, 9, 0])
        strategy = rng.choice([zlib.Z_DEFAULT_STRATEGY] * 4 + [zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED])
        co = zlib.compressobj(level, zlib.DEFLATED, -15, rng.choice([8, 9, 1]), strategy)
        # several deflate blocks inside one BGZF block now and then (Z_FULL_FLUSH / Z_SYNC_FLUSH between the pieces)
        cuts = sorted(rng.randrange(0, len(data) + 1) for _ in range(rng.choice([0, 0, 1, 3])))
        payload, prev = b"", 0
        for c in cuts:
            payload += co.compress(data[prev:c]) + co.flush(rng.choice([zlib.Z_FULL_FLUSH, zlib.Z_SYNC_FLUSH]))
            prev = c
        payload += co.compress(data[prev:]) + co.flush()
        if 18 + len(payload) + 8 > 65536:
            continue
        hdr = b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", 18 + len(payload) + 8 - 1)
        out.append((hdr + payload + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)), data))
    return out


for algo, tokens in ((0, 0), (0, 1), (1, 0)):  # workgroup kernel (commit decodes / commit from tokens), lane kernel
    ctx.set_option("inflate_algo", algo)
    ctx.set_option("inflate_tokens", tokens)
    ctx.set_option("inflate_lit_pairs", 1 - tokens)  # both forms of the symbol loop
    rng = orig(int(os.environ.get("INQ_SOAK_SEED", "4242")) + algo + 10 * tokens)
    n_blocks = 0
    for r in range(valid_rounds):
        items = _blocks(rng, 1500)
        comp = b"".join(b for b, _ in items)
        blocks = hipcall.scan_bgzf(comp)
        assert len(blocks) == len(items)
        rc, out, status = ctx.bgzf_inflate(comp, blocks, check=False)
        assert rc == 0 and not status.any(), (algo, r, [hex(int(s)) for s in status if s][:5])
        assert out.tobytes() == b"".join(d for _, d in items), (algo, r)
        n_blocks += len(items)
        print(f"  inflate_algo {algo} tokens {tokens} round {r}: {len(items)} blocks ok", flush=True)
    print(f"inflate_algo {algo} tokens {tokens}: {n_blocks} valid blocks of nine shapes inflate to their input", flush=True)
