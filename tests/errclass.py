"""Differential sweep over the reference's panic sites that depend on WHICH reads reach them.

`get_phase` runs for every record fetch() yields in phased mode (src/call.rs:349), `is_accidental_2d` only
from a soft-clip op inside `call_from_cigar`, i.e. only for reads that passed the filter (:303,357 -> :394).
Each case = eight good reads + one probe read at locus chr1:5000-5050 (window 4990..5060); the expected
outcome (row text or "the reference panics") comes from oracle/pyoracle.py run over the records.
"""
from __future__ import annotations

import itertools
from typing import Iterator, List, Optional, Tuple

from oracle import pyoracle as py
from tools import bamio

LOCUS = ("chr1", 5000, 5050)
REFS = [("chr1", 100_000)]

MAPQ = (5, 60)
HP = (None, ("C", 1), ("i", 1), ("s", 1))
# (pos, ref length, clip position): spanning / both ends inside the window / left end inside, right end beyond
GEOMETRY = {"spanning": (4800, 400), "inside": (5000, 30), "partial": (5000, 400)}
CLIP = (False, True)
SA = {
    "absent": None,
    "valid_2d": ("Z", "chr1,4900,-,300M,60,0;"),            # opposite strand, overlapping: accidental 2D
    "non_Z": ("i", 7),                                      # src/call.rs:429-432 panics
    "empty": ("Z", ""),                                     # sa_entries[0] out of bounds
    "semicolon": ("Z", ";"),
    "bad_pos": ("Z", "chr1,notanumber,-,50M,60,0;"),        # :450 unwrap on ParseIntError
    "bad_cigar": ("Z", "chr1,4900,-,M50,60,0;"),            # :469 unwrap on ParseIntError
    "three_fields": ("Z", "chr1,4900,-"),                   # :451 sa_entry[3] out of bounds
    "bad_pos_same_strand": ("Z", "chr1,notanumber,+,50M,60,0;"),  # returns false at :441-443, never parses POS
    "two_entries_garbage": ("Z", "x;y;"),                   # > 1 entry: false before anything is indexed
}


def good_reads() -> List[py.Record]:
    return [py.Record(pos=4800, cigar=[("M", 210), ("I", 12), ("M", 300)], hp=("C", 1 + k % 2), tid=0) for k in range(8)]


def cases() -> Iterator[Tuple[str, py.Record]]:
    for mapq, hp, geo, clip, sa in itertools.product(MAPQ, HP, GEOMETRY, CLIP, SA):
        pos, rlen = GEOMETRY[geo]
        cigar = ([("S", 20)] if clip else []) + [("M", rlen)]
        name = f"mapq{mapq}-hp{'none' if hp is None else hp[0]}-{geo}-{'clip' if clip else 'noclip'}-sa_{sa}"
        yield name, py.Record(pos=pos, cigar=cigar, mapq=mapq, flag=0, hp=hp, sa=SA[sa], tid=0)


def expected(recs: List[py.Record], unphased: bool, minlen: int = 5, support: int = 3) -> Optional[str]:
    """The row the reference prints, or None where it panics."""
    _, s, e = LOCUS
    try:
        if unphased:
            a, b, _tie = py.genotype_repeat_unphased(recs, 0, s, e, minlen, support)
        else:
            a, b = py.genotype_repeat_phased(recs, 0, s, e, minlen, support)
    except py.ReferencePanic:
        return None
    return py.format_row(LOCUS[0], s, e, a, b)


def write_bam(path: str, recs: List[py.Record]) -> List[py.Record]:
    """Coordinate-sorted BAM + .bai of the records; returns them in file order."""
    order = sorted(recs, key=lambda r: r.pos)
    w = bamio.BamWriter(path, REFS)
    for i, r in enumerate(order):
        tags = []
        if r.hp:
            tags.append(("HP", r.hp[0], r.hp[1]))
        if r.sa:
            tags.append(("SA", r.sa[0], r.sa[1]))
        w.add(f"r{i}", r.flag, 0, r.pos, r.mapq, r.cigar, tags)
    w.close()
    return order
