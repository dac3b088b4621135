"""What the reference DOES hold for the `call` path: an htslib-written index, `test-data/small-test.bam.bai` (kept as
tests/golden/reference_small-test.bam.bai; the BAM itself is a missing blob), and the one locus of `test-data/test.bed`
(chr7:154778571-154779363).  The product's index reading - where both front ends start (BaiIndex::scan_start) and where the
span planner stops (host/span_planner.cc) - is run on that file and checked against a computation written HERE from the SAM
specification (SAMv1 5.2: the BAI layout; 5.3: reg2bin / reg2bins and the linear index), not from the product's code.

No GPU involved.  The stub BAM carries the header only (195 contigs, chr7 = tid 6 with LN 159345973, src/call.rs:604): the
planner reads header + index, nothing else."""
import os
import shutil
import struct

import pytest

from inquistr_amd import call
from tools import bamio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
BAI = os.path.join(GOLDEN, "reference_small-test.bam.bai")
TID, START, END = 6, 154778571, 154779363  # test-data/test.bed:1 on the contig order of the index
PSEUDO_BIN = 37450


# ---------------------------------------------------------------- SAMv1 section 5.2: the file, read by hand
def parse_bai(path):
    d = open(path, "rb").read()
    assert d[:4] == b"BAI\1"
    (n_ref,) = struct.unpack_from("<i", d, 4)
    p = 8
    refs = []
    for _ in range(n_ref):
        (n_bin,) = struct.unpack_from("<i", d, p)
        p += 4
        bins = {}
        for _ in range(n_bin):
            b, n_chunk = struct.unpack_from("<Ii", d, p)
            p += 8
            bins[b] = [struct.unpack_from("<QQ", d, p + 16 * c) for c in range(n_chunk)]
            p += 16 * n_chunk
        (n_intv,) = struct.unpack_from("<i", d, p)
        p += 4
        ioff = list(struct.unpack_from("<%dQ" % n_intv, d, p))
        p += 8 * n_intv
        refs.append((bins, ioff))
    assert p in (len(d), len(d) - 8)  # optional n_no_coor
    return refs


# ---------------------------------------------------------------- SAMv1 section 5.3, transcribed from its C
def reg2bins(beg, end):
    end -= 1
    out = [0]
    for first, shift in ((1, 26), (9, 23), (73, 20), (585, 17), (4681, 14)):
        out += list(range(first + (beg >> shift), first + (end >> shift) + 1))
    return out


def bin_interval(b):
    """[start, end) of a bin id: level l has 8^l bins of 2^(29 - 3l) bases, ids from (8^l - 1) / 7."""
    for level in range(5, -1, -1):
        first = ((1 << (3 * level)) - 1) // 7
        if b >= first:
            size = 1 << (29 - 3 * level)
            return (b - first) * size, (b - first + 1) * size
    raise AssertionError(b)


def spec_query(bins, ioff, beg, end):
    """The region query of 5.3: chunks of the bins overlapping [beg, end) whose end lies behind the linear-index offset of
    beg's 16 kb window; sorted, neighbours merged."""
    min_off = ioff[beg >> 14] if (beg >> 14) < len(ioff) else 0
    chunks = sorted(c for b in reg2bins(beg, end) if b in bins and b != PSEUDO_BIN for c in bins[b] if c[1] > min_off)
    merged = []
    for u, v in chunks:
        if merged and u <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], v)
        else:
            merged.append([u, v])
    return min_off, chunks, merged


@pytest.fixture(scope="module")
def stub(tmp_path_factory):
    """A record-less BAM with the reference file's contig count (chr7 sixth, with its true length) and the reference's index."""
    d = tmp_path_factory.mktemp("refbai")
    refs = [(f"ref{t}", 250_000_000) for t in range(195)]
    refs[TID] = ("chr7", 159345973)
    bam = str(d / "small-test.bam")
    bamio.BamWriter(bam, refs).close(write_index=False)
    shutil.copy(BAI, bam + ".bai")
    return bam


def test_the_index_itself_is_what_the_spec_says():
    refs = parse_bai(BAI)
    assert len(refs) == 195
    assert [t for t, (bins, ioff) in enumerate(refs) if bins] == [TID]
    bins, ioff = refs[TID]
    assert bins[PSEUDO_BIN][1] == (8105, 0)  # mapped, unmapped
    real = {b: c for b, c in bins.items() if b != PSEUDO_BIN}
    # the file is coordinate-sorted: the linear index never decreases; every entry is the start of a record, i.e. lies inside
    # some chunk; and for a window records DO overlap (a leaf bin there, or a larger bin's record reaching in) the entry lies
    # in a chunk of a bin that overlaps the window
    assert all(v for v in ioff) and ioff == sorted(ioff)
    all_chunks = sorted(c for cs in real.values() for c in cs)
    for v in sorted(set(ioff)):
        assert any(u <= v < e for u, e in all_chunks), hex(v)
    for w, v in enumerate(ioff):
        if 4681 + w in real:
            cands = [c for b in reg2bins(w << 14, (w + 1) << 14) if b in real for c in real[b]]
            assert any(u <= v < e for u, e in cands), w
    # [3P] htslib fills windows no record overlaps from the RIGHT (hts_idx_finish): the 9 326 windows in front of the first
    # record (chr7:152.8 Mb) all carry the first record's offset - not 0, and not "the previous window's" (samtools 0.1)
    first_leaf = min(b for b in real if b >= 4681) - 4681
    first_rec = all_chunks[0][0]
    assert first_leaf > 9000 and set(ioff[: first_leaf - 64]) == {first_rec}


def test_scan_start_on_the_reference_index_equals_the_spec_linear_index():
    """Both front ends begin reading at BaiIndex::scan_start(tid, start - 10) (front_end.cc, span_planner.cc): on the
    reference's index that must be the linear-index offset of the window (5.3: "the smallest offset of an alignment that
    overlaps the window") - which is where htslib's own query starts to matter (chunks ending before it are dropped)."""
    L = call.load()
    bins, ioff = parse_bai(BAI)[TID]
    beg, end = START - 10, END + 10
    min_off, chunks, merged = spec_query(bins, ioff, beg, end)
    assert min_off > 0 and merged
    got = L.inq_host_bai_scan_start(BAI.encode(), TID, beg)
    assert got == min_off
    assert L.inq_host_bai_file_offset(BAI.encode(), TID, beg) == min_off >> 16
    # every chunk htslib would read for the locus ends behind that offset; the first one contains it or begins behind it
    assert all(v > min_off for _, v in merged)
    # other windows of the contig, also ones without records (htslib-written linear indexes carry the previous offset there,
    # other writers 0: the product looks forward for the next filled window)
    for pos in (0, 1 << 14, 100_000_000, START - 10 - (1 << 14), START + 5 * (1 << 14), (len(ioff) - 1) << 14):
        want = next((v for v in ioff[pos >> 14 :] if v), 0)
        assert L.inq_host_bai_scan_start(BAI.encode(), TID, pos) == want, pos
    assert L.inq_host_bai_scan_start(BAI.encode(), TID, len(ioff) << 14) == 0  # behind the last window: nothing there
    assert L.inq_host_bai_scan_start(BAI.encode(), 5, beg) == 0  # a contig without records


@pytest.mark.parametrize("how", ["region", "bed"])
def test_span_planner_on_the_reference_index(stub, how):
    """host/span_planner.cc on the reference's own locus: ONE span of ONE segment [vo_begin, vo_limit) with
      vo_begin = the linear-index offset of the window of start - 10, and
      vo_limit = the smallest chunk begin among the bins that start at or behind end + 10 (5.3's bin geometry: a bin's records
                 lie inside its interval, the file is sorted, so everything with pos < end + 10 precedes it),
    and, checked against the chunks of the index themselves: every record the spec's query could return for the locus lies
    inside that segment - bins wholly in front of end + 10 end before vo_limit, and no needed chunk ends before vo_begin."""
    bins, ioff = parse_bai(BAI)[TID]
    real = {b: c for b, c in bins.items() if b != PSEUDO_BIN}
    beg, end = START - 10, END + 10
    if how == "region":
        segs, tspan = call.plan_spans(stub, region=f"chr7:{START}-{END}")
    else:
        segs, tspan = call.plan_spans(stub, region_file=os.path.join(GOLDEN, "reference_test.bed"))
    assert len(segs) == 1 and segs[0][2] == 0 and tspan[0] == 0
    vo_begin, vo_limit, _ = segs[0]
    min_off, chunks, merged = spec_query(real, ioff, beg, end)
    assert vo_begin == min_off
    behind = [u for b, cs in real.items() if bin_interval(b)[0] >= end for u, _ in cs]
    assert behind and vo_limit == min(behind)
    assert vo_begin < vo_limit
    # bins that lie wholly in front of end + 10 hold only records with pos < end + 10: all of them precede the limit
    for b, cs in real.items():
        if bin_interval(b)[1] <= end:
            assert all(v <= vo_limit for _, v in cs), b
    # the 16 kb leaf bins inside the window are the tightest statement the index makes about the locus' records: every one
    # of their chunks that the linear index does not rule out lies inside the segment
    for b in reg2bins(beg, end):
        if b >= 4681 and b in real and bin_interval(b)[1] <= end:
            for u, v in real[b]:
                if v > min_off:
                    assert vo_begin <= max(u, min_off) and v <= vo_limit, (b, u, v)
    # the segment is a small part of the 73.7 MB file: 14 leaf windows around the locus, not the contig
    assert (vo_limit >> 16) - (vo_begin >> 16) < 4 << 20


def test_sweeping_the_contig_plans_monotone_disjoint_segments(stub, tmp_path):
    """A BED of 400 loci along chr7 on the reference's index: spans in file order, segments disjoint and ascending, every
    locus in exactly one span or in none (no record anywhere near), each begin / limit as the spec computation says."""
    bins, ioff = parse_bai(BAI)[TID]
    real = {b: c for b, c in bins.items() if b != PSEUDO_BIN}
    loci = [(154_000_000 + 5_000 * k, 154_000_000 + 5_000 * k + 300) for k in range(400)]
    bed = str(tmp_path / "sweep.bed")
    with open(bed, "w") as f:
        f.write("".join(f"chr7\t{s}\t{e}\n" for s, e in loci))
    segs, tspan = call.plan_spans(stub, region_file=bed, max_comp_bytes=1 << 20)
    assert segs and all(a[0] < a[1] for a in segs)
    assert all(x[1] <= y[0] or x[2] != y[2] for x, y in zip(segs, segs[1:]))
    assert [s[2] for s in segs] == sorted(s[2] for s in segs)
    starts_all = sorted(u for cs in real.values() for u, _ in cs)
    for i, (s, e) in enumerate(loci):
        w = (s - 10) >> 14
        want_begin = next((v for v in ioff[w:] if v), 0) if w < len(ioff) else 0
        behind = [u for b, cs in real.items() if bin_interval(b)[0] >= e + 10 for u, _ in cs]
        want_limit = min(behind) if behind else max(v for cs in real.values() for _, v in cs)
        if want_begin == 0 or want_limit <= want_begin:
            assert tspan[i] == 0xFFFFFFFF, i
            continue
        assert tspan[i] != 0xFFFFFFFF, i
        mine = [g for g in segs if g[2] == tspan[i]]
        assert any(g[0] <= want_begin and want_limit <= g[1] for g in mine), (i, want_begin, want_limit, mine)
    assert starts_all  # (the index has chunks at all)
