"""BAM + BED -> .inq text through libinquistr_host.so + the HIP library, against text built from
the naive Python restatement.  Mirrors the reference's own tests (src/call.rs:525-582:
test_region, test_region_bed, test_unphased), which only smoke the path; here the text is checked
byte for byte."""
import functools
import os
import subprocess

import numpy as np
import pytest

from inquistr_amd import call
from oracle import pyoracle as py
from tests.test_host_frontend import _expected, _make_case

pytestmark = pytest.mark.gpu


def _expected_text(loci, recs, unphased, minlen, support, sample, threads):
    p1, p2 = _expected(loci, recs, unphased, minlen, support)
    rows = [(c, s, e, a, b) for (c, s, e, _), a, b in zip(loci, p1, p2)]
    if threads > 1:  # src/call.rs:141 sort by (human chrom, start); equal keys keep BED order here
        rows.sort(key=functools.cmp_to_key(lambda x, y: py.human_compare(x[0], y[0]) or (x[1] > y[1]) - (x[1] < y[1])))
    return "\n".join([py.format_header(sample)] + [py.format_row(*r) for r in rows]) + "\n"


@pytest.mark.parametrize("seed,unphased,threads", [(11, False, 1), (12, True, 1), (13, False, 4), (14, True, 8)])
def test_genotype_repeats_text(tmp_path, seed, unphased, threads):
    minlen, support = 5, 3
    bam, bed, loci, recs = _make_case(tmp_path, seed, n_loci=80, ultra_long=(seed == 13))
    out = tmp_path / "out.inq"
    with open(out, "w") as f:
        call.genotype_repeats(bam, None, bed, minlen, support, threads, unphased, None, None, out=f)
    want = _expected_text(loci, recs, unphased, minlen, support, f"case{seed}.sorted", threads)
    assert open(out).read() == want


def test_cli_matches_library(tmp_path):
    bam, bed, loci, recs = _make_case(tmp_path, 21, n_loci=30)
    r = subprocess.run([call.CLI_PATH, "call", bam, "-R", bed, "--sample-name", "sample", "-t", "4"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == _expected_text(loci, recs, False, 5, 3, "sample", 4)
    # region string, unphased, custom minlen/support (src/call.rs:525-538 test_region shape)
    c, s, e, t = loci[0]
    r = subprocess.run([call.CLI_PATH, "call", bam, "-r", f"{c}:{s}-{e}", "-u", "-m", "3", "-s", "2"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == _expected_text([loci[0]], recs, True, 3, 2, "case21.sorted", 1)


def test_reference_bed_fixture_on_plumbing_bam(tmp_path):
    """BASELINE config #1 substitute: the reference's test-data/test.bed against a generated BAM whose
    header carries chr7 LN:159345973 (the only number the reference's tests assert, src/call.rs:604)."""
    import random

    from tests import gen
    from tools import bamio

    rng = random.Random(5)
    start, end = 154778571, 154779363
    recs = gen.random_locus_reads(rng, start, end, 40, long_every=6)
    recs.sort(key=lambda r: r.pos)
    bam = str(tmp_path / "plumbing.bam")
    w = bamio.BamWriter(bam, [("chr7", 159345973)])
    for i, r in enumerate(recs):
        tags = ([("HP", r.hp[0], r.hp[1])] if r.hp else []) + ([("SA", "Z", r.sa[1])] if r.sa else [])
        w.add(f"r{i}", r.flag, 0, r.pos, r.mapq, r.cigar, tags)
    w.close()
    bed = os.path.join(os.path.dirname(__file__), "golden", "reference_test.bed")
    for unphased in (False, True):
        out = tmp_path / f"o{unphased}.inq"
        with open(out, "w") as f:
            call.genotype_repeats(bam, None, bed, 5, 3, 4, unphased, "sample", None, out=f)
        want = _expected_text([("chr7", start, end, 0)], {0: recs}, unphased, 5, 3, "sample", 4)
        assert open(out).read() == want


@pytest.mark.parametrize("frontend", ["device", "host"])
def test_one_process_per_gpu_rehearsal(tmp_path, frontend):
    """`python -m torch.distributed.run ... -m inquistr_amd.call_dist` with two ranks (gloo, both on this box's one
    GPU): every rank runs the C++ driver on its slice, rank 0 gathers; the .inq equals the single-process CLI's."""
    import subprocess
    import sys

    from inquistr_amd import call
    from tools import make_synth_bam

    prefix = str(tmp_path / "w")
    make_synth_bam.write_native("phased10k", 4000, prefix)
    single = tmp_path / "single.inq"
    with open(single, "w") as f:
        call.genotype_repeats(prefix + ".bam", None, prefix + ".bed", 5, 3, 4, False, "S", None, out=f, frontend=frontend)
    out = tmp_path / "dist.inq"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29617", "-m", "inquistr_amd.call_dist", prefix + ".bam", "-R", prefix + ".bed", "-t", "4",
           "--sample-name", "S", "--backend", "gloo", "--same-device", "--frontend", frontend, "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out.read_text() == single.read_text() and out.read_text().count("\n") == 4001
