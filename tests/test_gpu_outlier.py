"""`inquiSTR outlier` on the GPU (inq_outlier_rows / inq_outlier) against the restatement, which the reference's
own unit tests pin (tests/test_outlier_oracle.py)."""
import gzip
import json
import os
import random
import subprocess

import numpy as np
import pytest

from inquistr_amd import call, hipcall
from oracle import outlier_oracle as oo

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    c = hipcall.Context(0)
    yield c
    c.close()


def test_reference_unit_test_vectors_on_the_gpu(ctx):
    for v in json.load(open(os.path.join(ROOT, "tests", "golden", "kat_outlier.json")))["vectors"]:
        vals = np.array([v["values"]], dtype=np.float32)
        rc, flags, keep = ctx.outlier_rows(vals, [vals.shape[1]], v["method"], minsize=0, zscore_cutoff=v.get("cutoff", 3.0),
                                           mincluster=v.get("mincluster", 1))
        assert rc == 0 and keep[0] == 1
        assert [oo._strip(v["samples"][k]) for k in np.nonzero(flags[0])[0]] == v["expect"], v["name"]


def _random_matrix(rng, n_rows, n_cols):
    rows, lens = [], []
    for _ in range(n_rows):
        n = n_cols if rng.random() < 0.8 else rng.randint(0, n_cols)
        kind = rng.random()
        if kind < 0.5:
            base = rng.choice([5, 12, 30, 200])
            r = [base + rng.choice([-1, 0, 0, 0, 1, 2]) + (0.5 if rng.random() < 0.2 else 0) for _ in range(n)]
            for _ in range(rng.choice([0, 1, 1, 2, 5])):
                if n:
                    r[rng.randrange(n)] = base * rng.choice([2, 3, 10]) + rng.randint(0, 400)
        elif kind < 0.7:
            r = [rng.choice([0, 1, 2, 3]) for _ in range(n)]  # mostly below minsize
        elif kind < 0.85:
            r = [rng.uniform(-50, 3000) for _ in range(n)]
        else:
            r = [float(rng.choice([7, 7, 7, 40])) for _ in range(n)]
        r = [float("nan") if rng.random() < 0.08 else x for x in r]
        if rng.random() < 0.05 and n:  # "inf" parses as an f32 too
            r[rng.randrange(n)] = rng.choice([float("inf"), float("-inf")])
        lens.append(n)
        rows.append(r + [0.0] * (n_cols - n))
    return np.array(rows, dtype=np.float32).reshape(n_rows, n_cols), np.array(lens, dtype=np.uint32)


@pytest.mark.parametrize("method", ["zscore", "dbscan"])
@pytest.mark.parametrize("seed,n_cols", [(1, 11), (2, 64), (3, 130), (4, 1), (5, 300)])
def test_outlier_rows_match_the_restatement(ctx, method, seed, n_cols):
    rng = random.Random(seed)
    vals, lens = _random_matrix(rng, 300, n_cols)
    minsize, cutoff, mincluster = rng.choice([0, 10, 10, 25]), rng.choice([1.0, 2.0, 3.0]), max(1, n_cols.bit_length() - 1)
    rc, flags, keep = ctx.outlier_rows(vals, lens, method, minsize=minsize, zscore_cutoff=cutoff, mincluster=mincluster)
    assert rc == 0
    for i in range(len(lens)):
        row = [np.float32(0) if np.isnan(x) else x for x in vals[i, : lens[i]]]
        if not row:
            assert keep[i] == 2
            continue
        if max(row) < np.float32(minsize):
            assert keep[i] == 0 and not flags[i].any()
            continue
        if method == "zscore":
            want = oo.z_score_flags(row, cutoff)
        else:
            try:
                want = oo.dbscan_flags(row, mincluster)
            except oo.ReferencePanic:
                assert keep[i] == 3
                continue
        assert keep[i] == 1
        assert list(flags[i, : lens[i]].astype(bool)) == want, (seed, i)
        assert not flags[i, lens[i]:].any()


@pytest.mark.parametrize("n_cols", [1, 3, 63, 64, 65, 127, 128, 129, 200, 255, 256, 257])
def test_zscore_through_the_lds_tile_equals_the_transposed_copy(ctx, n_cols):
    """Rows of at most 256 values take the LDS-tile kernel (one read of the row-major matrix, 64 or 32 rows per wave), wider ones and
    "outlier_tile" = 0 the transposed-copy kernel: the same flags and row states from both, on ragged rows (lengths 0 ... n_cols), a row
    count that is no multiple of a tile, NaNs, and equal to the plain-Python restatement."""
    rng = random.Random(1000 + n_cols)
    vals, lens = _random_matrix(rng, 777, n_cols)
    vals[rng.randrange(777), rng.randrange(n_cols)] = np.nan
    cutoff = rng.choice([1.0, 2.0, 3.0])
    rc, flags, keep = ctx.outlier_rows(vals, lens, "zscore", minsize=10, zscore_cutoff=cutoff, mincluster=3)
    ctx.set_option("outlier_tile", 0)
    try:
        rc2, flags2, keep2 = ctx.outlier_rows(vals, lens, "zscore", minsize=10, zscore_cutoff=cutoff, mincluster=3)
    finally:
        ctx.set_option("outlier_tile", 1)
    assert rc == rc2 == 0 and np.array_equal(flags, flags2) and np.array_equal(keep, keep2)
    for i in range(0, len(lens), 7):
        row = [np.float32(0) if np.isnan(x) else x for x in vals[i, : lens[i]]]
        if row and max(row) >= np.float32(10):
            assert keep[i] == 1 and list(flags[i, : lens[i]].astype(bool)) == oo.z_score_flags(row, cutoff), i
    assert keep.tolist().count(1) > 50


@pytest.mark.parametrize("method", ["zscore", "dbscan"])
def test_outlier_command_text(tmp_path, method):
    rng = random.Random(77)
    samples = [f"S{k}_H{h}" for k in range(20) for h in (1, 2)]
    vals, lens = _random_matrix(rng, 400, len(samples))
    lines = ["chromosome\tbegin\tend\t" + "\t".join(samples)]
    for i in range(len(lens)):
        if lens[i] == 0:
            lens[i] = 1  # a locus without values panics: tested separately
        cells = ["NaN" if np.isnan(x) else (str(int(x)) if float(x).is_integer() else repr(float(x))) for x in vals[i, : lens[i]]]
        lines.append(f"chr{1 + i % 3}\t{1000 * i}\t{1000 * i + 50}\t" + "\t".join(cells))
    plain, gz = tmp_path / "c.tsv", tmp_path / "c.tsv.gz"
    plain.write_text("\n".join(lines) + "\n")
    with gzip.open(gz, "wt") as f:
        f.write("\n".join(lines) + "\n")
    for path in (plain, gz):
        for kw in ({}, {"sample": "S3"}, {"minsize": 40, "zscore": 2.0}):
            out = tmp_path / "o.txt"
            with open(out, "w") as f:
                call.outlier(path, method=method, out=f, **kw)
            try:
                want = oo.outlier_text(lines, kw.get("minsize", 10), kw.get("zscore", 3.0), method,
                                       [kw["sample"]] if "sample" in kw else None)
            except oo.ReferencePanic:
                pytest.skip("a row without positive values under DBSCAN")
            assert out.read_text() == want, (path.name, kw)
    # the CLI prints the same bytes
    cli = os.path.join(ROOT, "inquistr_amd", "lib", "inquistr")
    r = subprocess.run([cli, "outlier", str(plain), "--method", method, "-z", "2.0", "--minsize", "40"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout == oo.outlier_text(lines, 40, 2.0, method)


def test_outlier_command_panics_like_the_reference(tmp_path):
    head = "chromosome\tbegin\tend\tA_H1\tA_H2\tB_H1\tB_H2"
    good = "chr1\t1\t9\t12\t12\t12\t99"

    def run(text, **kw):
        p = tmp_path / "c.tsv"
        p.write_text(text)
        out = tmp_path / "o.txt"
        with open(out, "w") as f:
            try:
                call.outlier(p, out=f, zscore=1.0, **kw)
                status = 0
            except call.CallError as e:
                status = e.status
        return status, out.read_text()

    assert run(head + "\n" + good + "\n") == (0, "chrom\tbegin\tend\toutliers\nchr1\t1\t9\tB\n")
    # the lines in front of a bad number are still reported, then the panic
    assert run(head + "\n" + good + "\nchr1\t20\t30\t1\tx\t3\t4\n" + good + "\n") == (101, "chrom\tbegin\tend\toutliers\nchr1\t1\t9\tB\n")
    assert run(head + "\n" + good + "\nchr1\t20\n")[0] == 101
    assert run(head + "\nchr1\t20\t30\n")[0] == 101  # no values: max of nothing
    assert run("chromosome\tbegin\tend\n" + good + "\n") == (101, "chrom\tbegin\tend\toutliers\n")
    assert run("")[0] == 101
    with pytest.raises(call.CallError) as e:
        call.outlier(tmp_path / "missing.tsv")
    assert e.value.status == 101
    (tmp_path / "subset.txt").write_text("A\n")
    with pytest.raises(call.CallError) as e:
        call.outlier(tmp_path / "c.tsv", sample="A", subset=tmp_path / "subset.txt")
    assert e.value.status == 101
