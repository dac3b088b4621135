"""pyoracle — a second, deliberately naive restatement of inquiSTR's `call` hot path.

TEST INFRASTRUCTURE ONLY.  Pure-Python loops, small cases only.  It exists so that the C
oracle (oracle/inq_oracle.c) has an independent second opinion: the two were written
separately from the reference's text and tests/ checks they agree on random inputs.

PARITY UNPINNED (see oracle/inq_oracle.h): nothing here was checked against the Rust binary.

Reference: wdecoster/inquiSTR v0.13.0, src/call.rs; line numbers are cited per function.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

OPS = "MIDNSHP=X"
U32 = 0xFFFFFFFF


class ReferencePanic(Exception):
    """The reference would panic here (unwrap / expect / index / explicit panic!)."""


@dataclass
class Record:
    """The slice of a BAM record the path looks at."""

    pos: int  # 0-based leftmost
    cigar: List[Tuple[str, int]]  # [(op_char, len)]
    mapq: int = 60
    flag: int = 0
    tid: int = 0
    hp: Optional[Tuple[str, int]] = None  # (aux type char, value) e.g. ("C", 1)
    sa: Optional[Tuple[str, object]] = None  # (aux type char, value) e.g. ("Z", "chr1,100,-,50M,60,0;")
    tags: dict = field(default_factory=dict)


def parse_cigar_string(s: str) -> List[Tuple[str, int]]:
    out, num = [], ""
    for ch in s:
        if ch.isdigit():
            num += ch
        else:
            out.append((ch, int(num)))
            num = ""
    return out


def reference_end(r: Record) -> int:
    """[3P] htslib bam_endpos, reached through Record::reference_end (src/call.rs:298,351,449)."""
    rlen = 0
    if not (r.flag & 0x4):
        rlen = sum(n for op, n in r.cigar if op in "MDN=X")
    if rlen == 0:
        rlen = 1
    return r.pos + rlen


def _rust_parse_i64(s: str) -> int:
    body = s[1:] if s[:1] in "+-" else s
    if not body or not all("0" <= c <= "9" for c in body):
        raise ReferencePanic(f"parse::<i64>({s!r})")
    v = int(s)
    if not -(1 << 63) <= v < (1 << 63):
        raise ReferencePanic("i64 overflow")
    return v


def cigar_to_rlen(cigar: str) -> int:
    """src/call.rs:461-477"""
    rlen, num = 0, ""
    for c in cigar:
        if "0" <= c <= "9":
            num += c
        else:
            n = _rust_parse_i64(num)
            if c in "M=XDN":
                rlen += n
            num = ""
    return rlen


def is_accidental_2d(r: Record) -> bool:
    """src/call.rs:415-459"""
    read_strand = "-" if r.flag & 0x10 else "+"
    if r.sa is None:
        return False
    typ, val = r.sa
    if typ != "Z":
        raise ReferencePanic("Unexpected type of Aux")
    entries = [x for x in val.split(";") if x != ""]
    if len(entries) > 1:
        return False
    if not entries:
        raise ReferencePanic("sa_entries[0]")
    f = entries[0].split(",")
    if len(f) < 3 or f[2] == "":
        raise ReferencePanic("sa_entry[2]")
    if read_strand == f[2][0]:
        return False
    start, end = r.pos, reference_end(r)
    sa_start = _rust_parse_i64(f[1])
    if len(f) < 4:
        raise ReferencePanic("sa_entry[3]")
    sa_end = sa_start + cigar_to_rlen(f[3])
    return max(start, sa_start) < min(end, sa_end)


def get_phase(r: Record) -> Optional[int]:
    """src/call.rs:482-491"""
    if r.hp is None:
        return None
    typ, v = r.hp
    if typ == "C":
        return v & 0xFF
    if typ == "i":
        return v & 0xFF  # `v as u8`
    raise ReferencePanic("Unexpected type of Aux")


def call_from_cigar(r: Record, minlen: int, start: int, end: int) -> Tuple[str, int]:
    """src/call.rs:377-413 -> ("Span"|"Clip", value)"""
    for op, _ in r.cigar:
        if op not in OPS:
            raise ReferencePanic("unknown cigar op")
    call = 0
    refpos = (r.pos + 1) & U32
    clipped = False
    for op, n in r.cigar:
        if op in "M=X":
            refpos = (refpos + n) & U32
        elif op == "D":
            if n > minlen and start < refpos < end:
                call -= n
            refpos = (refpos + n) & U32
        elif op == "S":
            if (not is_accidental_2d(r)) and n > minlen and start < refpos < end:
                call += n
                clipped = True
        elif op == "I":
            if n > minlen and start < refpos < end:
                call += n
        elif op == "N":
            refpos = (refpos + n) & U32
    return ("Clip" if clipped else "Span", call)


def median_str_length(calls: List[Tuple[str, int]], support: int) -> float:
    """src/call.rs:497-522"""
    if len(calls) < support:
        return math.nan
    spanning = [v for k, v in calls if k == "Span"]
    clipped = [v for k, v in calls if k == "Clip"]
    if len(spanning) <= support:
        clipped.sort(key=lambda k: -k)
        spanning.extend(clipped[0 : support - len(spanning)])
    spanning.sort()
    n = len(spanning)
    if n == 0:
        raise ReferencePanic("0/2 - 1 underflow")
    if n % 2 == 0:
        return float(spanning[n // 2 - 1] + spanning[n // 2]) / 2.0
    return float(spanning[n // 2])


def _fetch(recs: List[Record], tid: int, beg: int, end: int) -> List[Record]:
    """[3P] htslib region iterator: file order, tid match, pos < end, endpos > beg."""
    return [r for r in recs if r.tid == tid and r.pos < end and reference_end(r) > beg]


def genotype_repeat_phased(recs, tid, start, end, minlen, support):
    """src/call.rs:329-374"""
    if start < 10:
        raise ReferencePanic("start - 10 underflows u32")
    start_ext, end_ext = start - 10, end + 10
    calls = {1: [], 2: [], 0: []}
    for r in _fetch(recs, tid, start_ext, end_ext):
        phase = get_phase(r)
        if (
            phase is None
            or (start_ext < (r.pos & U32) and (reference_end(r) & U32) < end_ext)
            or r.mapq <= 10
        ):
            continue
        call = call_from_cigar(r, minlen, start_ext, end_ext)
        if phase not in calls:
            raise ReferencePanic("calls.get_mut(&phase).unwrap()")
        calls[phase].append(call)
    return median_str_length(calls[1], support), median_str_length(calls[2], support)


def genotype_repeat_unphased(recs, tid, start, end, minlen, support):
    """src/call.rs:279-327; returns (phase1, phase2, tie)"""
    if start < 10:
        raise ReferencePanic("start - 10 underflows u32")
    start_ext, end_ext = start - 10, end + 10
    calls = []
    for r in _fetch(recs, tid, start_ext, end_ext):
        if start_ext < (r.pos & U32) or (reference_end(r) & U32) < end_ext or r.mapq <= 10:
            continue
        calls.append(call_from_cigar(r, minlen, start_ext, end_ext))
    calls = sorted(calls, key=lambda c: c[1])  # stable: equal values stay in file order
    k = len(calls) // 2
    tie = False
    if 1 <= k < len(calls) and calls[k - 1][1] == calls[k][1]:
        kinds = {c[0] for c in calls if c[1] == calls[k][1]}
        tie = len(kinds) == 2
    return median_str_length(calls[:k], support), median_str_length(calls[k:], support), tie


def format_f64(v: float) -> str:
    """[3P] Rust `{}` for f64: the shortest digits that read back as v (Python's repr finds the same), written positionally - never
    an exponent: 2.0 ** 60 is "1152921504606847000" -, "NaN", "inf", "-0"."""
    from decimal import Decimal

    v = float(v)
    if math.isnan(v):
        return "NaN"
    if math.isinf(v):
        return "-inf" if v < 0 else "inf"
    r = repr(v)
    if "e" in r or "E" in r:
        r = format(Decimal(r), "f")
    if r.endswith(".0"):
        r = r[:-2]
    return r


def format_row(chrom, start, end, p1, p2) -> str:
    """src/call.rs:57-65"""
    return f"{chrom}\t{start}\t{end}\t{format_f64(p1)}\t{format_f64(p2)}"


def format_header(sample: str) -> str:
    """src/call.rs:101"""
    return f"chromosome\tbegin\tend\t{sample}_H1\t{sample}_H2"


def sample_name(bam_path: str) -> str:
    """src/call.rs:91-100"""
    name = bam_path.rstrip("/").split("/")[-1]
    if name != ".." and "." in name[1:]:
        name = name[: name.rindex(".")]
    return name.replace(".bam", "").replace(".cram", "")


def human_compare(a: str, b: str) -> int:
    """[3P] human-sort 0.2.2 compare, as used by Genotype::cmp (src/call.rs:33-38)."""
    i = j = 0
    while i < len(a) and j < len(b):
        if a[i].isdigit() and b[j].isdigit():
            i2 = i
            while i2 < len(a) and a[i2].isdigit():
                i2 += 1
            j2 = j
            while j2 < len(b) and b[j2].isdigit():
                j2 += 1
            x, y = int(a[i:i2]), int(b[j:j2])
            if x != y:
                return -1 if x < y else 1
            i, j = i2, j2
        else:
            if a[i] != b[j]:
                return -1 if a[i] < b[j] else 1
            i += 1
            j += 1
    if i < len(a):
        return 1
    if j < len(b):
        return -1
    return 0
