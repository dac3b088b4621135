"""`combine` (src/combine.rs:27-59) on the reference's own fixtures (test-data/file{1,2,3}.inq[.gz], kept
under tests/golden/ as data).  The reference's tests (src/combine.rs:62-78) only smoke it; the expected
text here is the rule restated in three lines of Python."""
import gzip
import os
import subprocess

import pytest

from inquistr_amd import call

G = os.path.join(os.path.dirname(__file__), "golden")


def _lines(p):
    op = gzip.open if p.endswith(".gz") else open
    with op(p, "rt") as f:
        return [l.rstrip("\n").rstrip("\r") for l in f]


def _expected(files):
    cols = [_lines(f) for f in files]
    return "".join("\t".join([l] + [x for o in cols[1:] for x in o[i].split("\t")[3:]]) + "\n" for i, l in enumerate(cols[0]))


@pytest.mark.parametrize("ext", ["", ".gz"])
def test_reference_fixtures(tmp_path, ext):
    files = [os.path.join(G, f"reference_file{i}.inq{ext}") for i in (1, 2, 3)]
    out = tmp_path / "c.tsv"
    with open(out, "w") as f:
        call.combine(files, out=f)
    text = open(out).read()
    assert text == _expected(files)
    assert text.splitlines()[2] == "chr1\t10627\t10997\t150.0\t117.0\t150.0\t117.0\tNaN\t117.0"
    r = subprocess.run([call.CLI_PATH, "combine"] + files, capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout == text


def test_mixed_and_single(tmp_path):
    files = [os.path.join(G, "reference_file1.inq"), os.path.join(G, "reference_file3.inq.gz")]
    out = tmp_path / "c.tsv"
    with open(out, "w") as f:
        call.combine(files, out=f)
    assert open(out).read() == _expected(files)
    with open(out, "w") as f:
        call.combine(files[:1], out=f)
    assert open(out).read() == "".join(l + "\n" for l in _lines(files[0]))


def test_panics(tmp_path):
    short = tmp_path / "short.inq"
    short.write_text("chr1\t1\t2\t3\t4\n")
    with pytest.raises(call.CallError) as e:  # file2.next().unwrap() on None, src/combine.rs:49
        with open(tmp_path / "o", "w") as f:
            call.combine([os.path.join(G, "reference_file1.inq"), str(short)], out=f)
    assert e.value.status == 101
    with pytest.raises(call.CallError) as e:  # src/combine.rs:30-32
        with open(tmp_path / "o", "w") as f:
            call.combine([str(tmp_path / "missing.inq")], out=f)
    assert e.value.status == 101 and "does not exist" in e.value.message
    # a longer second file is simply not read to the end
    longer = tmp_path / "long.inq"
    longer.write_text("".join(f"c\t1\t2\t{i}\t{i}\n" for i in range(9)))
    with open(tmp_path / "o", "w") as f:
        call.combine([os.path.join(G, "reference_file1.inq"), str(longer)], out=f)
    assert len(open(tmp_path / "o").read().splitlines()) == 5
