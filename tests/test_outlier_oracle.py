"""`inquiSTR outlier` restatement (oracle/outlier_oracle.py) against the reference's own unit-test vectors and
hand-derived ones; host-side text rules that need no GPU."""
import json
import os

import numpy as np
import pytest

from oracle import outlier_oracle as oo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def kat():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "kat_outlier.json")))["vectors"]


def test_oracle_reproduces_the_reference_unit_tests(kat):
    assert sum(1 for v in kat if v["name"].startswith("reference")) == 2
    for v in kat:
        vals = [np.float32(x) for x in v["values"]]
        if v["method"] == "zscore":
            got = oo.z_score_outliers(vals, v["samples"], v["cutoff"])
        else:
            got = oo.dbscan_outliers(vals, v["samples"], v["mincluster"])
        assert got == v["expect"], v["name"]


def test_parse_f32_follows_rust():
    for ok in ["1", "-1.5", "+2", ".5", "5.", "1e3", "1E-2", "NaN", "nan", "inf", "-Infinity"]:
        oo.parse_f32(ok)
    for bad in ["", " 1", "1 ", "e5", ".", "0x10", "1_0", "1,5", "--1"]:
        with pytest.raises(oo.ReferencePanic):
            oo.parse_f32(bad)


def test_outlier_text_rules():
    head = "chromosome\tbegin\tend\tA_H1\tA_H2\tB_H1\tB_H2\tC_H1\tC_H2\tD_H1\tD_H2"
    rows = ["chr1\t10\t20\t1\t2\t2\t3\t1\tNaN\t3\t500", "chr1\t30\t40\t1\t2\t2\t3\t1\t5\t3\t2", "chr2\t5\t9\t400\t2\t2\t3\t1\t5\t3\t2"]
    assert oo.outlier_text([head] + rows, 10, 2.0) == "chrom\tbegin\tend\toutliers\nchr1\t10\t20\tD\nchr2\t5\t9\tA\n"
    assert oo.outlier_text([head] + rows, 10, 2.0, subset=["A"]) == "chrom\tbegin\tend\toutliers\nchr2\t5\t9\tA\n"
    assert oo.outlier_text([head] + rows, 1000, 2.0) == "chrom\tbegin\tend\toutliers\n"  # nothing reaches minsize
    with pytest.raises(oo.ReferencePanic):
        oo.outlier_text([head, "chr1\t1\t2\tx"], 10, 2.0)
    with pytest.raises(oo.ReferencePanic):
        oo.outlier_text(["chromosome\tbegin\tend"], 10, 2.0)


def test_outlier_command_without_gpu(tmp_path):
    """No CPU path for the arithmetic: without a gfx950 device the command ends with status 1; the reference's
    argument panics come first."""
    import torch

    from inquistr_amd import call

    p = tmp_path / "c.tsv"
    p.write_text("chromosome\tbegin\tend\tA_H1\tA_H2\nchr1\t1\t9\t12\t99\n")
    with pytest.raises(call.CallError) as e:
        call.outlier(tmp_path / "missing.tsv")
    assert e.value.status == 101 and "does not exist" in e.value.message
    (tmp_path / "s.txt").write_text("A\n")
    with pytest.raises(call.CallError) as e:
        call.outlier(p, sample="A", subset=tmp_path / "s.txt")
    assert e.value.status == 101
    if not torch.cuda.is_available():
        with pytest.raises(call.CallError) as e, open(tmp_path / "o.txt", "w") as f:
            call.outlier(p, out=f)
        assert e.value.status == 1 and "no CPU fallback" in e.value.message


@pytest.mark.parametrize("method", ["zscore", "dbscan"])
def test_c_restatement_agrees_with_the_python_one(method):
    import random

    rng = random.Random(5)
    n_rows, n_cols = 200, 37
    vals = np.zeros((n_rows, n_cols), dtype=np.float32)
    lens = np.zeros(n_rows, dtype=np.uint32)
    for i in range(n_rows):
        n = rng.choice([0, 1, 5, n_cols, n_cols])
        base = rng.choice([3, 12, 40])
        row = [base + rng.choice([-1, 0, 0, 1, 0.5]) for _ in range(n)]
        for _ in range(rng.choice([0, 1, 3])):
            if n:
                row[rng.randrange(n)] = rng.choice([base * 7.0, float("nan"), float("inf"), -4.0])
        vals[i, :n] = row
        lens[i] = n
    flags, keep = oo.c_outlier_rows(vals, lens, method, minsize=10, cutoff=2.0, mincluster=5, threads=2)
    for i in range(n_rows):
        row = [np.float32(0) if np.isnan(x) else x for x in vals[i, : lens[i]]]
        if not row:
            assert keep[i] == 2
        elif max(row) < np.float32(10):
            assert keep[i] == 0
        else:
            try:
                want = oo.z_score_flags(row, 2.0) if method == "zscore" else oo.dbscan_flags(row, 5)
            except oo.ReferencePanic:
                assert keep[i] == 3
                continue
            assert keep[i] == 1 and list(flags[i, : lens[i]].astype(bool)) == want, i
