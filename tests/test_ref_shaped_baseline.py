"""The CPU baseline program (oracle/ref_shaped_call: reference-shaped control flow around the oracle,
index fetch per locus) against the naive Python restatement, all three modes."""
import os
import subprocess

import pytest

from tests.test_gpu_end_to_end import _expected_text
from tests.test_host_frontend import _make_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "ref_shaped_call")


@pytest.fixture(scope="module")
def exe():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref_shaped_call"], stdout=subprocess.DEVNULL)
    return EXE


@pytest.mark.parametrize("mode,threads,unphased", [("C", 1, False), ("B", 4, True), ("A", 3, False)])
def test_modes_match_python(tmp_path, exe, mode, threads, unphased):
    bam, bed, loci, recs = _make_case(tmp_path, 31, n_loci=50, ultra_long=True)
    r = subprocess.run([exe, bam, bed, mode, str(threads), str(int(unphased)), "5", "3", "S"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == _expected_text(loci, recs, unphased, 5, 3, "S", 1 if mode == "C" else threads)
