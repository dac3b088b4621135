"""ctypes loader for the C oracle (oracle/liborcinq.so).  TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np

from inquistr_amd.batch import Batch, InqBatchC, InqResultC, Result

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liborcinq.so")


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("inq_oracle.c", "inq_oracle.h")]
    src.append(os.path.join(_HERE, "..", "include", "inquistr_hip.h"))
    stale = not os.path.exists(_LIB_PATH) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liborcinq.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class OrcCall(C.Structure):
    _fields_ = [("value", C.c_int64), ("clipped", C.c_int)]


class OrcRecord(C.Structure):
    _fields_ = [
        ("tid", C.c_int32),
        ("pos", C.c_int64),
        ("flag", C.c_uint16),
        ("mapq", C.c_uint8),
        ("n_cigar", C.c_uint32),
        ("cigar", C.POINTER(C.c_uint32)),
        ("hp_type", C.c_char),
        ("hp_value", C.c_int64),
        ("sa_type", C.c_char),
        ("sa", C.c_char_p),
        ("is2d_given", C.c_int),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_bam_endpos.restype = C.c_int64
        L.orc_bam_endpos.argtypes = [C.POINTER(OrcRecord)]
        L.orc_cigar_to_rlen.restype = C.c_int64
        L.orc_cigar_to_rlen.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
        L.orc_is_accidental_2d.restype = C.c_int
        L.orc_is_accidental_2d.argtypes = [C.POINTER(OrcRecord), C.POINTER(C.c_int)]
        L.orc_get_phase.restype = C.c_int
        L.orc_get_phase.argtypes = [C.POINTER(OrcRecord), C.POINTER(C.c_uint8), C.POINTER(C.c_int)]
        L.orc_call_from_cigar.restype = OrcCall
        L.orc_call_from_cigar.argtypes = [C.POINTER(OrcRecord), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_int)]
        L.orc_median_str_length.restype = C.c_double
        L.orc_median_str_length.argtypes = [C.POINTER(OrcCall), C.c_size_t, C.c_size_t, C.POINTER(C.c_int)]
        L.orc_genotype_repeat_phased.restype = C.c_int
        L.orc_genotype_repeat_phased.argtypes = [
            C.POINTER(OrcRecord), C.c_size_t, C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_size_t,
            C.POINTER(C.c_double), C.POINTER(C.c_double),
        ]
        L.orc_genotype_repeat_unphased.restype = C.c_int
        L.orc_genotype_repeat_unphased.argtypes = [
            C.POINTER(OrcRecord), C.c_size_t, C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_size_t,
            C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int),
        ]
        L.orc_call_batch.restype = C.c_int
        L.orc_call_batch.argtypes = [C.POINTER(InqBatchC), C.POINTER(InqResultC), C.c_int]
        L.orc_format_f64.restype = C.c_size_t
        L.orc_format_f64.argtypes = [C.c_double, C.c_char_p, C.c_size_t]
        L.orc_format_row.restype = C.c_size_t
        L.orc_format_row.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_char_p, C.c_size_t]
        L.orc_format_header.restype = C.c_size_t
        L.orc_format_header.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.orc_sample_name.restype = C.c_size_t
        L.orc_sample_name.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.orc_human_compare.restype = C.c_int
        L.orc_human_compare.argtypes = [C.c_char_p, C.c_char_p]
        L.orc_parse_region.restype = C.c_int
        L.orc_parse_region.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_check_interval.restype = C.c_int
        L.orc_check_interval.argtypes = [C.c_uint32, C.c_uint32, C.c_int64]
        _lib = L
    return _lib


OPS = "MIDNSHP=X"


class Rec:
    """Keeps the buffers an OrcRecord points to alive."""

    def __init__(self, pos, cigar, mapq=60, flag=0, tid=0, hp=None, sa=None, is2d_given=-1):
        if cigar and isinstance(cigar[0], tuple):
            words = [(n << 4) | OPS.index(o) for o, n in cigar]
        else:
            words = list(cigar)
        self._cig = (C.c_uint32 * max(1, len(words)))(*words)
        self._sa = None
        r = OrcRecord()
        r.tid, r.pos, r.flag, r.mapq = tid, pos, flag, mapq
        r.n_cigar = len(words)
        r.cigar = C.cast(self._cig, C.POINTER(C.c_uint32))
        if hp is not None:
            r.hp_type, r.hp_value = hp[0].encode(), hp[1]
        else:
            r.hp_type = b"\0"
        if sa is not None:
            r.sa_type = sa[0].encode()
            if sa[0] == "Z":
                self._sa = sa[1].encode()
                r.sa = self._sa
        else:
            r.sa_type = b"\0"
        r.is2d_given = is2d_given
        self.c = r


def bam_endpos(rec: Rec) -> int:
    return lib().orc_bam_endpos(C.byref(rec.c))


def cigar_to_rlen(s: str) -> Tuple[int, int]:
    p = C.c_int(0)
    v = lib().orc_cigar_to_rlen(s.encode(), C.byref(p))
    return v, p.value


def is_accidental_2d(rec: Rec) -> Tuple[bool, int]:
    p = C.c_int(0)
    v = lib().orc_is_accidental_2d(C.byref(rec.c), C.byref(p))
    return bool(v), p.value


def get_phase(rec: Rec) -> Tuple[Optional[int], int]:
    p = C.c_int(0)
    ph = C.c_uint8(0)
    has = lib().orc_get_phase(C.byref(rec.c), C.byref(ph), C.byref(p))
    return (ph.value if has else None), p.value


def call_from_cigar(rec: Rec, minlen: int, start: int, end: int) -> Tuple[str, int, int]:
    p = C.c_int(0)
    c = lib().orc_call_from_cigar(C.byref(rec.c), minlen, start, end, C.byref(p))
    return ("Clip" if c.clipped else "Span"), c.value, p.value


def median_str_length(calls: Sequence[Tuple[str, int]], support: int) -> Tuple[float, int]:
    n = len(calls)
    arr = (OrcCall * max(1, n))()
    for i, (k, v) in enumerate(calls):
        arr[i].value, arr[i].clipped = v, 1 if k == "Clip" else 0
    p = C.c_int(0)
    v = lib().orc_median_str_length(arr, n, support, C.byref(p))
    return v, p.value


def _rec_array(recs: List[Rec]):
    arr = (OrcRecord * max(1, len(recs)))()
    for i, r in enumerate(recs):
        arr[i] = r.c
    return arr


def genotype_repeat_phased(recs: List[Rec], tid, start, end, minlen, support):
    a, b = C.c_double(), C.c_double()
    p = lib().orc_genotype_repeat_phased(_rec_array(recs), len(recs), tid, start, end, minlen, support, C.byref(a), C.byref(b))
    return a.value, b.value, p


def genotype_repeat_unphased(recs: List[Rec], tid, start, end, minlen, support):
    a, b, t = C.c_double(), C.c_double(), C.c_int(0)
    p = lib().orc_genotype_repeat_unphased(
        _rec_array(recs), len(recs), tid, start, end, minlen, support, C.byref(a), C.byref(b), C.byref(t)
    )
    return a.value, b.value, bool(t.value), p


def call_batch(batch: Batch, debug: bool = False, threads: int = 1) -> Tuple[int, Result]:
    res = Result.alloc(batch, debug=debug)
    bc, rc = batch.as_c(), res.as_c()
    code = lib().orc_call_batch(C.byref(bc), C.byref(rc), threads)
    res.n_tie_loci = int(rc.n_tie_loci)
    return code, res


def call_batch_raw(bc: InqBatchC, rc: InqResultC, threads: int = 1) -> int:
    return lib().orc_call_batch(C.byref(bc), C.byref(rc), threads)


def format_f64(v: float) -> str:
    buf = C.create_string_buffer(512)  # (f64::MAX takes 309 digits, the smallest subnormal 326 characters)
    lib().orc_format_f64(v, buf, 512)
    return buf.value.decode()


def format_row(chrom, start, end, p1, p2) -> str:
    buf = C.create_string_buffer(512)
    lib().orc_format_row(chrom.encode(), start, end, p1, p2, buf, 512)
    return buf.value.decode()


def format_header(sample: str) -> str:
    buf = C.create_string_buffer(1024)
    lib().orc_format_header(sample.encode(), buf, 1024)
    return buf.value.decode()


def sample_name(path: str) -> str:
    buf = C.create_string_buffer(1024)
    lib().orc_sample_name(path.encode(), buf, 1024)
    return buf.value.decode()


def human_compare(a: str, b: str) -> int:
    return lib().orc_human_compare(a.encode(), b.encode())


def parse_region(reg: str):
    buf = C.create_string_buffer(512)
    s, e = C.c_uint32(), C.c_uint32()
    p = lib().orc_parse_region(reg.encode(), buf, 512, C.byref(s), C.byref(e))
    return buf.value.decode(), s.value, e.value, p


def check_interval(start, end, chrom_len) -> int:
    return lib().orc_check_interval(start, end, chrom_len)
