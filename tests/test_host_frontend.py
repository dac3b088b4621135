"""Host side without a GPU: BGZF/BAM/BAI reader, targets, text, and the sweep front end.

The front end's batches go through the CPU oracle (allowed in tests) and the result is compared
with the naive Python restatement run over ALL records of the BAM per locus — i.e. against what
`bam.fetch()` + the per-locus genotyper would see.
"""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

from inquistr_amd import call
from oracle import pyoracle as py
from tests import gen
from tools import bamio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def test_host_library_exports_every_declared_symbol():
    import re

    hdr = open(os.path.join(ROOT, "include", "inquistr_host.h")).read()
    declared = set(re.findall(r"\b(inq_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(call.HOST_ABI_SYMBOLS)
    L = call.load()
    for s in declared:
        assert hasattr(L, s), s


def test_reference_bai_fixture():
    """The reference's own index (test-data/small-test.bam.bai; the BAM itself is not distributed):
    195 contigs, data on chr7 (tid 6) only, 8105 mapped reads — facts of SURVEY.md."""
    L = call.load()
    n, m, u, nb, ni = C.c_uint32(), C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    p = os.path.join(GOLDEN, "reference_small-test.bam.bai").encode()
    assert L.inq_host_bai_stats(p, C.byref(n), 6, C.byref(m), C.byref(u), C.byref(nb), C.byref(ni)) == 0
    assert (n.value, m.value, u.value) == (195, 8105, 0) and nb.value > 0 and ni.value > 0
    assert L.inq_host_bai_stats(p, C.byref(n), 5, C.byref(m), C.byref(u), C.byref(nb), C.byref(ni)) == 0
    assert nb.value == 0


def test_text_side_matches_kats(kat):
    L = call.load()
    buf = C.create_string_buffer(1024)
    for v, s in kat["format_f64"]:
        L.inq_host_format_f64(float("nan") if v == "NaN" else float(v), buf, 1024)
        assert buf.value.decode() == s
    for v in kat["format_row"]:
        f = lambda x: float("nan") if x == "NaN" else float(x)
        L.inq_host_format_row(v["chrom"].encode(), v["start"], v["end"], f(v["p1"]), f(v["p2"]), buf, 1024)
        assert buf.value.decode() == v["expect"]
    for s, h in kat["header"]:
        L.inq_host_format_header(s.encode(), buf, 1024)
        assert buf.value.decode() == h
    for p, s in kat["sample_name"]:
        L.inq_host_sample_name(p.encode(), buf, 1024)
        assert buf.value.decode() == s, p
    for a, b, c in kat["human_compare"]:
        assert L.inq_host_human_compare(a.encode(), b.encode()) == c
        assert L.inq_host_human_compare(b.encode(), a.encode()) == -c
    for v in kat["parse_region"]:
        s, e = C.c_uint32(), C.c_uint32()
        rc = L.inq_host_parse_region(v["reg"].encode(), b"chr7", 159345973, buf, 1024, C.byref(s), C.byref(e))
        if v.get("panic"):
            assert rc == 101, v
        else:
            assert rc == 0 and [buf.value.decode(), s.value, e.value] == v["expect"]
    for v in kat["check_interval"]:
        s, e = C.c_uint32(), C.c_uint32()
        reg = f"c:{v['start']}-{v['end']}".encode()
        rc = L.inq_host_parse_region(reg, b"c" if v["len"] >= 0 else b"other", max(v["len"], 0), buf, 1024, C.byref(s), C.byref(e))
        assert (rc == 0) == v["ok"], v


REFS = [("chr1", 400_000), ("chr2", 300_000), ("chr10", 250_000), ("chrX", 200_000), ("chrEmpty", 50_000)]


def _make_case(tmp_path, seed, n_loci=60, ultra_long=False, block=bamio.BLOCK):
    """Random coordinate-sorted BAM + BED.  Returns (bam, bed, loci, records_by_tid)."""
    rng = random.Random(seed)
    loci = []
    recs = {t: [] for t in range(len(REFS))}
    for _ in range(n_loci):
        t = rng.choice([0, 0, 1, 2, 3, 4])
        ln = REFS[t][1]
        start = rng.randint(2000, ln - 4000)
        end = start + rng.randint(0, 250)
        loci.append((REFS[t][0], start, end, t))
        if t == 4:
            continue  # a contig with loci but no reads
        for r in gen.random_locus_reads(rng, start, end, rng.choice([0, 2, 5, 9, 30]), long_every=7):
            recs[t].append(r)
    # nested / overlapping loci and a far-away pair on chr2 (gap jump)
    loci.append(("chr1", 100_000, 100_400, 0))
    loci.append(("chr1", 100_050, 100_060, 0))
    loci.append(("chr2", 10_000, 10_020, 1))
    loci.append(("chr2", 290_000, 290_030, 1))
    for t, s, e in ((0, 100_000, 100_400), (1, 10_000, 10_020), (1, 290_000, 290_030)):
        recs[t] += gen.random_locus_reads(rng, s, e, 12, long_every=5)
    if ultra_long:  # one read spanning > 64 kb across several loci
        recs[1].append(py.Record(pos=5_000, cigar=[("M", 290_000)], mapq=60, hp=("C", 1)))
    # background reads nowhere near a locus, unmapped-but-placed reads, and unplaced reads at the end
    for t in (0, 1):
        for _ in range(40):
            recs[t].append(py.Record(pos=rng.randint(0, REFS[t][1] - 2000), cigar=gen.random_cigar(rng, 9), mapq=60,
                                     hp=("C", rng.choice([1, 2]))))
    bam = str(tmp_path / f"case{seed}.sorted.bam")
    w = bamio.BamWriter(bam, REFS, block=block)
    k = 0
    for t in range(len(REFS)):
        recs[t].sort(key=lambda r: r.pos)  # stable: file order among equal positions = insertion order
        for r in recs[t]:
            tags = []
            if r.hp:
                tags.append(("HP", r.hp[0], r.hp[1]))
            if r.sa:
                tags.append(("SA", "Z", r.sa[1]))
            tags.append(("NM", "i", 3))
            if k % 3 == 0:  # methylation-style tags before/after the ones the path reads: B arrays, Z strings
                tags.insert(0, ("ML", "B", ("C", [1, 2, 250] * (k % 5))))
                tags.append(("MM", "Z", "C+m,5,12,0;"))
                tags.append(("qs", "f", 12.5))
                tags.append(("ts", "A", "+"))
            r.tid = t
            w.add(f"read{k}", r.flag, t, r.pos, r.mapq, r.cigar, tags, l_seq=rng.choice([0, 0, 7]))
            k += 1
    for _ in range(5):
        w.add(f"unplaced{k}", 4, -1, -1, 0, [], [], l_seq=4)
        k += 1
    w.close()
    rng.shuffle(loci)
    bed = str(tmp_path / f"case{seed}.bed")
    with open(bed, "w") as f:
        f.write("# comment line\n")
        for c, s, e, _ in loci:
            f.write(f"{c}\t{s}\t{e}\n")
        f.write("\n")
    return bam, bed, loci, recs


def _expected(loci, recs, unphased, minlen, support):
    p1, p2 = [], []
    for _, s, e, t in loci:
        if unphased:
            a, b, _tie = py.genotype_repeat_unphased(recs[t], t, s, e, minlen, support)
        else:
            a, b = py.genotype_repeat_phased(recs[t], t, s, e, minlen, support)
        p1.append(a)
        p2.append(b)
    return np.array(p1), np.array(p2)


@pytest.mark.parametrize("seed,unphased,threads,words", [(1, False, 1, 0), (2, True, 1, 2000), (3, False, 4, 600), (4, True, 3, 0)])
def test_frontend_batches_reproduce_per_locus_fetch(tmp_path, orc, seed, unphased, threads, words):
    minlen, support = 5, [3, 1, 2, 3][seed % 4]
    bam, bed, loci, recs = _make_case(tmp_path, seed, ultra_long=(seed == 3))
    fe = call.FrontEnd(bam, region_file=bed, minlen=minlen, support=support, threads=threads, unphased=unphased,
                       max_batch_words=words)
    assert fe.sample == f"case{seed}.sorted"
    assert fe.targets() == [(c, s, e) for c, s, e, _ in loci]
    got1 = np.full(len(loci), -12345.0)
    got2 = np.full(len(loci), -12345.0)
    n_batches = 0
    for batch, idx in fe.batches():
        n_batches += 1
        code, res = orc.call_batch(batch)
        assert code == 0
        got1[idx], got2[idx] = res.phase1, res.phase2
        # every read is stored once per batch and pads to 16 bytes
        assert batch.cigar.shape[0] % 4 == 0
    if words:
        assert n_batches > 4
    want1, want2 = _expected(loci, recs, unphased, minlen, support)
    assert gen.same_f64(got1, want1) and gen.same_f64(got2, want2)
    fe.close()


def test_region_string_and_long_cigar_tag(tmp_path, orc):
    """-r with one locus; one read carries 70 000 ops (real CIGAR in CG:B,I behind <l_seq>S<ref>N)."""
    rng = random.Random(9)
    big = gen.random_cigar(rng, 70_000)
    span = py.reference_end(py.Record(pos=0, cigar=big))
    start = 1000 + span // 2
    recs = [py.Record(pos=1000, cigar=big, mapq=60, hp=("C", 1), tid=0) for _ in range(3)]
    recs += [py.Record(pos=start - 200, cigar=[("M", 150), ("I", 30 + k), ("M", 400)], hp=("i", 2), tid=0) for k in range(3)]
    recs.sort(key=lambda r: r.pos)
    bam = str(tmp_path / "long.bam")
    w = bamio.BamWriter(bam, [("chr7", span + 100_000)])
    for i, r in enumerate(recs):
        w.add(f"r{i}", 0, 0, r.pos, 60, r.cigar, [("HP", r.hp[0], r.hp[1])], l_seq=5)
    w.close()
    fe = call.FrontEnd(bam, region=f"chr7:{start}-{start + 100}", sample_name="S")
    (batch, idx), = list(fe.batches())
    assert batch.reads["n_cigar"].max() == 70_000
    code, res = orc.call_batch(batch)
    a, b = py.genotype_repeat_phased(recs, 0, start, start + 100, 5, 3)
    assert code == 0 and gen.same_f64(res.phase1, np.array([a])) and gen.same_f64(res.phase2, np.array([b]))
    assert fe.sample == "S"


def test_reference_error_behaviour(tmp_path):
    bam, bed, loci, _ = _make_case(tmp_path, 5, n_loci=5)
    with pytest.raises(call.CallError) as e:  # src/call.rs:87-90: exit(1)
        call.FrontEnd(str(tmp_path / "missing.bam"), region="chr1:100-200")
    assert e.value.status == 1 and "is not valid" in e.value.message
    with pytest.raises(call.CallError) as e:  # :197-200: neither -r nor -R
        call.FrontEnd(bam)
    assert e.value.status == 1
    with pytest.raises(call.CallError) as e:  # both given
        call.FrontEnd(bam, region="chr1:100-200", region_file=bed)
    assert e.value.status == 1
    with pytest.raises(call.CallError) as e:  # test_region_wrong_chromosome, src/call.rs:584-598
        call.FrontEnd(bam, region="7:154778571-154779363")
    assert e.value.status == 101 and "not in the fasta file" in e.value.message
    with pytest.raises(call.CallError) as e:  # end >= LN (src/repeats.rs:108-114)
        call.FrontEnd(bam, region="chr1:100-400000")
    assert e.value.status == 101
    with pytest.raises(call.CallError) as e:  # end < start (src/repeats.rs:102-104)
        call.FrontEnd(bam, region="chr1:200-100")
    assert e.value.status == 101 and "smaller than start" in e.value.message
    with pytest.raises(call.CallError) as e:  # start < 10: u32 underflow -> fetch fails -> expect panics
        call.FrontEnd(bam, region="chr1:5-100")
    assert e.value.status == 101
    with pytest.raises(call.CallError) as e:  # no comma stripping in repeats.rs:13-29
        call.FrontEnd(bam, region="chr1:1,000-2,000")
    assert e.value.status == 101
    noidx = str(tmp_path / "noidx.bam")
    w = bamio.BamWriter(noidx, REFS)
    w.close(write_index=False)
    with pytest.raises(call.CallError) as e:  # IndexedReader::from_path without an index panics (:242-243)
        call.FrontEnd(noidx, region="chr1:100-200")
    assert e.value.status == 101 and "Error opening local BAM" in e.value.message
    cram = str(tmp_path / "x.cram")
    open(cram, "wb").write(b"CRAM")
    with pytest.raises(call.CallError) as e:
        call.FrontEnd(cram, region="chr1:100-200")
    assert e.value.status == 1 and "CRAM" in e.value.message


def test_chrom_lengths_from_bam_header(tmp_path):
    """src/call.rs:600-605 asserts chr7 LN == 159345973 on its (undistributed) BAM; here the same number sits in a
    generated header and is observed through RepeatInterval's `end < LN` rule (src/repeats.rs:108-114)."""
    bam = str(tmp_path / "plumbing.bam")
    w = bamio.BamWriter(bam, [("chr7", 159345973)])
    w.add("r", 0, 0, 154778000, 60, [("M", 2000)], [("HP", "C", 1)])
    w.close()
    assert call.FrontEnd(bam, region="chr7:154778571-154779363").targets() == [("chr7", 154778571, 154779363)]
    assert call.FrontEnd(bam, region="chr7:10-159345972").targets() == [("chr7", 10, 159345972)]
    with pytest.raises(call.CallError) as e:
        call.FrontEnd(bam, region="chr7:10-159345973")
    assert e.value.status == 101
    bed = os.path.join(GOLDEN, "reference_test.bed")  # the reference's test-data/test.bed
    assert call.FrontEnd(bam, region_file=bed).targets() == [("chr7", 154778571, 154779363)]


def test_bed_reader_rules(tmp_path):
    bam, _, _, _ = _make_case(tmp_path, 6, n_loci=3)

    def targets(text):
        p = str(tmp_path / "t.bed")
        open(p, "w").write(text)
        return call.FrontEnd(bam, region_file=p).targets()

    assert targets("chr1\t100\t200\n#c\n\nchr2\t300\t400\r\n") == [("chr1", 100, 200), ("chr2", 300, 400)]
    assert targets("chr1\t100\t200\tname\t0\t+\nchr2\t300\t400\tn2\t1\t-\n") == [("chr1", 100, 200), ("chr2", 300, 400)]
    for bad in ("chr1\t100\t200\nchr2\t300\t400\textra\n",  # csv: records of unequal length
                "chr1\t100\n",                               # fewer than 3 fields
                "chr1\t1e3\t2000\n",                          # u64 parse
                "chr1 100 200\n",                             # not tab separated
                "chr1\t100\t5000000000\n",                    # > u32
                "track name=x\nchr1\t100\t200\n"):
        with pytest.raises(call.CallError) as e:
            targets(bad)
        assert e.value.status == 101, bad


def test_hp_of_other_type_panics_only_when_phased(tmp_path):
    bam = str(tmp_path / "hp.bam")
    w = bamio.BamWriter(bam, [("chr1", 100_000)])
    w.add("a", 0, 0, 900, 60, [("M", 300)], [("HP", "s", 1)])
    w.add("far", 0, 0, 50_000, 60, [("M", 300)], [("HP", "S", 1)])  # never fetched: no panic from it
    w.close()
    fe = call.FrontEnd(bam, region="chr1:1010-1090", unphased=True)
    assert len(list(fe.batches())) == 1
    fe = call.FrontEnd(bam, region="chr1:1010-1090", unphased=False)
    with pytest.raises(call.CallError) as e:
        list(fe.batches())
    assert e.value.status == 101 and "Aux" in e.value.message
    fe = call.FrontEnd(bam, region="chr1:3010-3090", unphased=False)  # no record fetched at all
    assert len(list(fe.batches())) == 1


def test_unsorted_bam_is_refused(tmp_path):
    bam = str(tmp_path / "unsorted.bam")
    w = bamio.BamWriter(bam, [("chr1", 100_000)])
    w.add("a", 0, 0, 5_000, 60, [("M", 300)], [("HP", "C", 1)])
    w.add("b", 0, 0, 900, 60, [("M", 300)], [("HP", "C", 1)])  # goes backwards
    w.close()
    fe = call.FrontEnd(bam, region="chr1:1010-6000")
    with pytest.raises(call.CallError) as e:
        list(fe.batches())
    assert "not coordinate-sorted" in e.value.message


def test_cli_exit_codes(tmp_path):
    exe = call.CLI_PATH
    r = subprocess.run([exe, "call", str(tmp_path / "nope.bam"), "-r", "chr1:100-200"], capture_output=True, text=True)
    assert r.returncode == 1 and "is not valid" in r.stderr and r.stdout == ""
    bam, bed, _, _ = _make_case(tmp_path, 7, n_loci=3)
    r = subprocess.run([exe, "call", bam, "-r", "7:154778571-154779363"], capture_output=True, text=True)
    assert r.returncode == 101 and "panicked" in r.stderr and r.stdout == ""
    r = subprocess.run([exe, "call"], capture_output=True, text=True)
    assert r.returncode == 2 and "Usage" in r.stderr
    r = subprocess.run([exe, "call", bam, "--bogus"], capture_output=True, text=True)
    assert r.returncode == 2
