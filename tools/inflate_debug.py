#!/usr/bin/env python3
"""Per-block diagnosis of the GPU inflate against zlib: status bits and the first differing byte of every block."""
import os, random, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inquistr_amd import hipcall
from tests.test_gpu_front import _bgzf, _payloads

ctx = hipcall.Context(0)
ctx.set_option("inflate_algo", int(os.environ.get("ALGO", "0")))
rng = random.Random(5)
n_bad = 0
only = os.environ.get("ONLY")
for pi, data in enumerate(_payloads(rng)):
    if only is not None and pi != int(only):
        continue
    for level, strategy in [(0, 0), (1, 0), (6, 0), (9, 0), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE), (9, zlib.Z_FILTERED)]:
        d = data[:65000] if level == 0 and len(data) > 65000 else data
        try:
            blob = _bgzf(d, level, strategy)
        except AssertionError:
            continue
        blocks = hipcall.scan_bgzf(blob)
        rc, out, status = ctx.bgzf_inflate(blob, blocks, check=False)
        got = out.tobytes()
        if rc != 0 or got != d:
            n_bad += 1
            diff = next((i for i in range(min(len(got), len(d))) if got[i] != d[i]), -1)
            print(f"payload {pi} len {len(d)} level {level} strategy {strategy}: rc {rc} status {[hex(int(s)) for s in status]} first diff at {diff} comp {len(blob)}")
print("bad:", n_bad)
