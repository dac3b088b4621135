#!/usr/bin/env python3
"""End-to-end soak (GPU box): random BAM + BED through the product (C++ sweep front end + HIP kernels) against
text built from the naive Python restatement over every record of the BAM.  Run by hand.
usage: python tools/soak_e2e.py [--cases 60] [--frontend host|device] [--seed0 5000]"""
import argparse, os, pathlib, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inquistr_amd import call
from tests.test_gpu_end_to_end import _expected_text
from tests.test_host_frontend import _make_case

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=60)
ap.add_argument("--frontend", default=None, choices=[None, "host", "device"])
ap.add_argument("--seed0", type=int, default=5000)
a = ap.parse_args()
bad = 0
with tempfile.TemporaryDirectory() as td:
    for i in range(a.cases):
        seed = a.seed0 + i
        unphased, threads = bool(i & 1), [1, 4, 7][i % 3]
        if a.frontend == "device":  # vary how the file is cut into spans and segments, and the BGZF block size
            os.environ["INQ_SPAN_GAP_BYTES"] = str([0, 128 << 10, 2000][i % 3])
            os.environ["INQ_SPAN_MB"] = "0" if i % 2 else "1"
        bam, bed, loci, recs = _make_case(pathlib.Path(td), seed, n_loci=30 + (i % 5) * 25, ultra_long=(i % 4 == 0),
                                          block=[0xFF00, 1500, 9000][i % 3] if a.frontend == "device" else 0xFF00)
        out = os.path.join(td, "o.inq")
        with open(out, "w") as f:
            call.genotype_repeats(bam, None, bed, 5, [3, 1, 2][i % 3], threads, unphased, None, None, out=f, frontend=a.frontend)
        want = _expected_text(loci, recs, unphased, 5, [3, 1, 2][i % 3], f"case{seed}.sorted", threads)
        if open(out).read() != want:
            bad += 1
            print(f"MISMATCH seed={seed} unphased={unphased} threads={threads}", flush=True)
        for p in (bam, bam + ".bai", bed):
            os.unlink(p)
        if i % 10 == 9:
            print(f"  {i + 1} cases, {bad} mismatches so far", flush=True)
print(f"e2e soak done ({a.frontend or 'auto'} front end): {a.cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
