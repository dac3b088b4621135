"""Both oracles (C and Python) against the hand-derived known-answer vectors.

Mirrors the spirit of the reference's inline tests (src/call.rs:524-628), which only smoke
the path; the numeric expectations here are derived by hand from the reference's code
(tests/golden/kat_call.json) because no reference test asserts a number on this path.
"""
import math

import pytest

from oracle import pyoracle as py


def _num(x):
    return math.nan if x == "NaN" else float(x)


def _same(a, b):
    return (math.isnan(a) and math.isnan(b)) or a == b


def _pyrec(d):
    return py.Record(
        pos=d["pos"],
        cigar=py.parse_cigar_string(d["cigar"]),
        mapq=d.get("mapq", 60),
        flag=d.get("flag", 0),
        hp=tuple(d["hp"]) if d.get("hp") else None,
        sa=("Z", d["sa"]) if d.get("sa") else None,
    )


def _crec(orc, d):
    return orc.Rec(
        pos=d["pos"],
        cigar=py.parse_cigar_string(d["cigar"]),
        mapq=d.get("mapq", 60),
        flag=d.get("flag", 0),
        hp=tuple(d["hp"]) if d.get("hp") else None,
        sa=("Z", d["sa"]) if d.get("sa") else None,
    )


def test_call_from_cigar(kat, orc):
    for v in kat["call_from_cigar"]:
        kind, val = v["expect"]
        got_py = py.call_from_cigar(_pyrec(v), v["minlen"], v["start"], v["end"])
        assert got_py == (kind, val), v["name"]
        k, x, panic = orc.call_from_cigar(_crec(orc, v), v["minlen"], v["start"], v["end"])
        assert (k, x, panic) == (kind, val, 0), v["name"]


def test_cigar_to_rlen(kat, orc):
    for v in kat["cigar_to_rlen"]:
        if v.get("panic"):
            with pytest.raises(py.ReferencePanic):
                py.cigar_to_rlen(v["cigar"])
            assert orc.cigar_to_rlen(v["cigar"])[1] != 0
        else:
            assert py.cigar_to_rlen(v["cigar"]) == v["expect"]
            assert orc.cigar_to_rlen(v["cigar"]) == (v["expect"], 0)


def test_bam_endpos(kat, orc):
    for v in kat["bam_endpos"]:
        assert py.reference_end(_pyrec(v)) == v["expect"]
        assert orc.bam_endpos(_crec(orc, v)) == v["expect"]


def test_get_phase(kat, orc):
    for v in kat["get_phase"]:
        d = {"pos": 0, "cigar": "10M", "hp": v["hp"]}
        if v.get("panic"):
            with pytest.raises(py.ReferencePanic):
                py.get_phase(_pyrec(d))
            assert orc.get_phase(_crec(orc, d))[1] != 0
        else:
            assert py.get_phase(_pyrec(d)) == v["expect"]
            assert orc.get_phase(_crec(orc, d)) == (v["expect"], 0)


def test_median_str_length(kat, orc):
    for v in kat["median_str_length"]:
        calls = [tuple(c) for c in v["calls"]]
        want = _num(v["expect"])
        assert _same(py.median_str_length(calls, v["support"]), want), v
        got, panic = orc.median_str_length(calls, v["support"])
        assert panic == 0 and _same(got, want), v


def test_median_support_zero_panics(orc):
    with pytest.raises(py.ReferencePanic):
        py.median_str_length([], 0)
    assert orc.median_str_length([], 0)[1] != 0


def test_loci(kat, orc):
    for v in kat["loci"]:
        want = [_num(x) for x in v["expect"]]
        pyrecs = [_pyrec(r) for r in v["reads"]]
        crecs = [_crec(orc, r) for r in v["reads"]]
        if v["mode"] == "phased":
            a, b = py.genotype_repeat_phased(pyrecs, 0, v["start"], v["end"], v["minlen"], v["support"])
            ca, cb, panic = orc.genotype_repeat_phased(crecs, 0, v["start"], v["end"], v["minlen"], v["support"])
            assert panic == 0
        else:
            a, b, tie = py.genotype_repeat_unphased(pyrecs, 0, v["start"], v["end"], v["minlen"], v["support"])
            ca, cb, ctie, panic = orc.genotype_repeat_unphased(
                crecs, 0, v["start"], v["end"], v["minlen"], v["support"]
            )
            assert panic == 0
            assert tie == ctie == bool(v.get("tie", False)), v["name"]
        assert _same(a, want[0]) and _same(b, want[1]), (v["name"], a, b)
        assert _same(ca, want[0]) and _same(cb, want[1]), (v["name"], ca, cb)


def test_phase_out_of_range_panics(orc):
    d = {"pos": 900, "cigar": "300M", "hp": ["C", 3]}
    with pytest.raises(py.ReferencePanic):
        py.genotype_repeat_phased([_pyrec(d)], 0, 1010, 1090, 5, 3)
    assert orc.genotype_repeat_phased([_crec(orc, d)], 0, 1010, 1090, 5, 3)[2] != 0
    # a read the filter drops never reaches the HashMap lookup (src/call.rs:350-358)
    d2 = dict(d, mapq=5)
    a, b = py.genotype_repeat_phased([_pyrec(d2)], 0, 1010, 1090, 5, 3)
    assert math.isnan(a) and math.isnan(b)
    assert orc.genotype_repeat_phased([_crec(orc, d2)], 0, 1010, 1090, 5, 3)[2] == 0


def test_start_below_ten_is_outside_domain(orc):
    d = {"pos": 0, "cigar": "300M", "hp": ["C", 1]}
    with pytest.raises(py.ReferencePanic):
        py.genotype_repeat_phased([_pyrec(d)], 0, 9, 90, 5, 3)
    assert orc.genotype_repeat_phased([_crec(orc, d)], 0, 9, 90, 5, 3)[2] != 0


def test_formatting(kat, orc):
    for v, s in kat["format_f64"]:
        assert py.format_f64(_num(v)) == s
        assert orc.format_f64(_num(v)) == s
    for v in kat["format_row"]:
        args = (v["chrom"], v["start"], v["end"], _num(v["p1"]), _num(v["p2"]))
        assert py.format_row(*args) == v["expect"]
        assert orc.format_row(*args) == v["expect"]
    for s, h in kat["header"]:
        assert py.format_header(s) == h and orc.format_header(s) == h
    for p, s in kat["sample_name"]:
        assert py.sample_name(p) == s, p
        assert orc.sample_name(p) == s, p


def test_human_compare(kat, orc):
    for a, b, c in kat["human_compare"]:
        assert py.human_compare(a, b) == c and py.human_compare(b, a) == -c
        assert orc.human_compare(a, b) == c and orc.human_compare(b, a) == -c


def test_region_and_interval(kat, orc):
    for v in kat["parse_region"]:
        chrom, s, e, panic = orc.parse_region(v["reg"])
        if v.get("panic"):
            assert panic != 0, v
        else:
            assert panic == 0 and [chrom, s, e] == v["expect"]
    for v in kat["check_interval"]:
        assert (orc.check_interval(v["start"], v["end"], v["len"]) == 0) == v["ok"], v


def test_reference_bed_fixture_is_one_locus():
    # the reference's own BED fixture (test-data/test.bed:1), kept as data in tests/golden/
    import os

    p = os.path.join(os.path.dirname(__file__), "golden", "reference_test.bed")
    rows = [l.rstrip("\n").split("\t") for l in open(p) if l.strip()]
    assert rows == [["chr7", "154778571", "154779363"]]
