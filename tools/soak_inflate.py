#!/usr/bin/env python3
"""GPU box: the inflate fuzz of tests/test_gpu_front.py with other seeds (accept / reject and bytes vs zlib).
usage: python tools/soak_inflate.py [rounds]"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_gpu_front as t  # noqa: E402
from inquistr_amd import hipcall  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ctx = hipcall.Context(0)
orig = random.Random
for r in range(rounds):
    seed = 9000 + r

    class Seeded(orig):  # the test builds its own Random(2024): give it another stream each round
        def __init__(self, _ignored=None):
            super().__init__(seed)

    random.Random = Seeded
    try:
        t.test_inflate_fuzz_agrees_with_zlib_on_mutated_streams(ctx)
    finally:
        random.Random = orig
    print(f"round {r}: 4000 damaged streams agree with zlib", flush=True)
print(f"inflate soak done: {rounds} rounds, 0 disagreements")
