"""Host-side batch layout for the C ABI in include/inquistr_hip.h.

`Batch` holds the numpy buffers an `inq_batch_t` points to and packs decoded reads into
them; nothing here computes on CIGAR words (that is the GPU's job) beyond copying them.

Reference counterpart: the data `genotype_repeat_{phased,unphased}` pulls out of
rust-htslib records one locus at a time (src/call.rs:279-374); here it is laid out once
per batch so the device can stream it.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

# ---- ctypes mirror of include/inquistr_hip.h ------------------------------------------

INQ_OK = 0
INQ_ERR_ARG = -1
INQ_ERR_SUPPORT_ZERO = -2
INQ_ERR_PHASE = -3
INQ_ERR_CIGAR_OP = -4
INQ_ERR_LOCUS = -5
INQ_ERR_RANGE = -6
INQ_ERR_INDEX = -7
INQ_ERR_HIP = -8
INQ_ERR_NOMEM = -9
INQ_ERR_NO_DEVICE = -10
INQ_ERR_INFLATE = -11
INQ_ERR_BAM = -12
INQ_ERR_AUX = -13

INQ_READ_UNMAPPED = 0x01
INQ_READ_REVERSE = 0x02
INQ_READ_HAS_HP = 0x04
INQ_READ_IS_2D = 0x08
INQ_READ_SA_PANIC = 0x10  # has an S op and is_accidental_2d would panic: raised for kept reads only

INQ_PAIR_CLIP = 0x01
INQ_PAIR_FETCHED = 0x02
INQ_PAIR_KEPT = 0x04

READ_DTYPE = np.dtype(
    [
        ("cigar_off4", "<u4"),
        ("n_cigar", "<u4"),
        ("pos", "<i4"),
        ("mapq", "u1"),
        ("bits", "u1"),
        ("phase", "u1"),
        ("reserved", "u1"),
    ]
)
assert READ_DTYPE.itemsize == 16

OP_CODE = {c: i for i, c in enumerate("MIDNSHP=X")}


class InqBatchC(C.Structure):
    _fields_ = [
        ("n_reads", C.c_uint64),
        ("n_cigar_words", C.c_uint64),
        ("n_pairs", C.c_uint64),
        ("n_loci", C.c_uint64),
        ("cigar", C.c_void_p),
        ("reads", C.c_void_p),
        ("pair_read", C.c_void_p),
        ("locus_pair_off", C.c_void_p),
        ("locus_start", C.c_void_p),
        ("locus_end", C.c_void_p),
        ("minlen", C.c_uint32),
        ("support", C.c_uint32),
        ("unphased", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class InqResultC(C.Structure):
    _fields_ = [
        ("phase1", C.c_void_p),
        ("phase2", C.c_void_p),
        ("pair_call", C.c_void_p),
        ("pair_bits", C.c_void_p),
        ("n_tie_loci", C.c_uint64),
    ]


def _ptr(a: Optional[np.ndarray]) -> Optional[int]:
    if a is None or a.size == 0:
        # a valid non-NULL address keeps "empty" distinct from "missing"
        return None if a is None else a.ctypes.data or None
    return a.ctypes.data


@dataclass
class Batch:
    """One batch of loci with the reads offered to each (host numpy buffers)."""

    cigar: np.ndarray  # u32 [n_cigar_words], reads padded to 4 words with 0
    reads: np.ndarray  # READ_DTYPE [n_reads]
    pair_read: np.ndarray  # u32 [n_pairs]
    locus_pair_off: np.ndarray  # u64 [n_loci + 1]
    locus_start: np.ndarray  # u32 [n_loci]
    locus_end: np.ndarray  # u32 [n_loci]
    minlen: int = 5
    support: int = 3
    unphased: bool = False

    @property
    def n_loci(self) -> int:
        return int(self.locus_start.shape[0])

    @property
    def n_pairs(self) -> int:
        return int(self.pair_read.shape[0])

    @property
    def n_reads(self) -> int:
        return int(self.reads.shape[0])

    def as_c(self) -> InqBatchC:
        """ctypes view; the Batch must outlive the returned struct."""
        for name, dt in (
            ("cigar", np.uint32),
            ("pair_read", np.uint32),
            ("locus_pair_off", np.uint64),
            ("locus_start", np.uint32),
            ("locus_end", np.uint32),
        ):
            a = getattr(self, name)
            if a.dtype != dt or not a.flags["C_CONTIGUOUS"]:
                setattr(self, name, np.ascontiguousarray(a, dtype=dt))
        if self.reads.dtype != READ_DTYPE or not self.reads.flags["C_CONTIGUOUS"]:
            self.reads = np.ascontiguousarray(self.reads, dtype=READ_DTYPE)
        b = InqBatchC()
        b.n_reads = self.n_reads
        b.n_cigar_words = int(self.cigar.shape[0])
        b.n_pairs = self.n_pairs
        b.n_loci = self.n_loci
        b.cigar = _ptr(self.cigar)
        b.reads = _ptr(self.reads)
        b.pair_read = _ptr(self.pair_read)
        b.locus_pair_off = _ptr(self.locus_pair_off)
        b.locus_start = _ptr(self.locus_start)
        b.locus_end = _ptr(self.locus_end)
        b.minlen = int(self.minlen)
        b.support = int(self.support)
        b.unphased = 1 if self.unphased else 0
        b.reserved = 0
        return b

    def cigar_ops_per_pair(self) -> np.ndarray:
        return self.reads["n_cigar"][self.pair_read].astype(np.int64)

    def algorithmic_bytes(self) -> int:
        """HBM bytes one pass must move (DESIGN.md "Algorithmic bytes"): every CIGAR word of
        every pair once, one 16-B read descriptor + one 4-B index per pair, 16 B of locus
        input (8-B offset, start, end) and 16 B of output per locus."""
        ops = int(self.cigar_ops_per_pair().sum())
        return 4 * ops + 20 * self.n_pairs + 32 * self.n_loci

    def slice_loci(self, lo: int, hi: int) -> "Batch":
        """Loci [lo, hi) with only the reads they reference (used for sharding and sampling)."""
        p0, p1 = int(self.locus_pair_off[lo]), int(self.locus_pair_off[hi])
        pr = self.pair_read[p0:p1]
        uniq, inv = np.unique(pr, return_inverse=True)
        reads = self.reads[uniq].copy()
        n4 = (reads["n_cigar"].astype(np.int64) + 3) // 4
        new_off4 = np.zeros(len(reads), dtype=np.int64)
        if len(reads):
            new_off4[1:] = np.cumsum(n4)[:-1]
        total4 = int(n4.sum())
        cig = np.zeros(total4 * 4, dtype=np.uint32)
        for k in range(len(reads)):
            src = int(reads["cigar_off4"][k]) * 4
            n = int(reads["n_cigar"][k])
            dst = int(new_off4[k]) * 4
            cig[dst : dst + n] = self.cigar[src : src + n]
        reads["cigar_off4"] = new_off4.astype(np.uint32)
        return Batch(
            cigar=cig,
            reads=reads,
            pair_read=inv.astype(np.uint32),
            locus_pair_off=(self.locus_pair_off[lo : hi + 1] - np.uint64(p0)).astype(np.uint64),
            locus_start=self.locus_start[lo:hi].copy(),
            locus_end=self.locus_end[lo:hi].copy(),
            minlen=self.minlen,
            support=self.support,
            unphased=self.unphased,
        )


@dataclass
class Result:
    phase1: np.ndarray
    phase2: np.ndarray
    pair_call: Optional[np.ndarray] = None
    pair_bits: Optional[np.ndarray] = None
    n_tie_loci: int = 0

    @classmethod
    def alloc(cls, batch: Batch, debug: bool = False) -> "Result":
        return cls(
            phase1=np.full(batch.n_loci, np.nan, dtype=np.float64),
            phase2=np.full(batch.n_loci, np.nan, dtype=np.float64),
            pair_call=np.zeros(batch.n_pairs, dtype=np.int64) if debug else None,
            pair_bits=np.zeros(batch.n_pairs, dtype=np.uint8) if debug else None,
        )

    def as_c(self) -> InqResultC:
        r = InqResultC()
        r.phase1 = _ptr(self.phase1)
        r.phase2 = _ptr(self.phase2)
        r.pair_call = _ptr(self.pair_call)
        r.pair_bits = _ptr(self.pair_bits)
        r.n_tie_loci = 0
        return r


# ---- packing decoded reads ------------------------------------------------------------


def encode_cigar(ops: Sequence[Tuple[str, int]]) -> np.ndarray:
    """[(op_char, len)] -> BAM-native u32 words."""
    return np.array([(n << 4) | OP_CODE[o] for o, n in ops], dtype=np.uint32)


class BatchBuilder:
    """Accumulates reads and (locus -> reads) lists, then lays them out as a Batch."""

    def __init__(self, minlen: int = 5, support: int = 3, unphased: bool = False):
        self.minlen, self.support, self.unphased = minlen, support, unphased
        self._cig: List[np.ndarray] = []
        self._reads: List[tuple] = []
        self._off4 = 0
        self._loci: List[Tuple[int, int, List[int]]] = []

    def add_read(
        self,
        pos: int,
        cigar_words: np.ndarray,
        mapq: int = 60,
        phase: Optional[int] = None,
        reverse: bool = False,
        unmapped: bool = False,
        is_2d: bool = False,
        sa_panic: bool = False,
    ) -> int:
        w = np.asarray(cigar_words, dtype=np.uint32)
        n = int(w.shape[0])
        pad = (-n) % 4
        self._cig.append(w)
        if pad:
            self._cig.append(np.zeros(pad, dtype=np.uint32))
        bits = (
            (INQ_READ_UNMAPPED if unmapped else 0)
            | (INQ_READ_REVERSE if reverse else 0)
            | (INQ_READ_HAS_HP if phase is not None else 0)
            | (INQ_READ_IS_2D if is_2d else 0)
            | (INQ_READ_SA_PANIC if sa_panic else 0)
        )
        self._reads.append((self._off4, n, pos, mapq, bits, (phase or 0) & 0xFF, 0))
        self._off4 += (n + pad) // 4
        return len(self._reads) - 1

    def add_locus(self, start: int, end: int, read_indices: Iterable[int]) -> int:
        self._loci.append((start, end, list(read_indices)))
        return len(self._loci) - 1

    def build(self) -> Batch:
        cig = np.concatenate(self._cig) if self._cig else np.zeros(0, dtype=np.uint32)
        reads = np.array(self._reads, dtype=READ_DTYPE) if self._reads else np.zeros(0, dtype=READ_DTYPE)
        offs = np.zeros(len(self._loci) + 1, dtype=np.uint64)
        pr: List[int] = []
        for j, (_, _, lst) in enumerate(self._loci):
            pr.extend(lst)
            offs[j + 1] = len(pr)
        return Batch(
            cigar=cig.astype(np.uint32),
            reads=reads,
            pair_read=np.array(pr, dtype=np.uint32),
            locus_pair_off=offs,
            locus_start=np.array([l[0] for l in self._loci], dtype=np.uint32),
            locus_end=np.array([l[1] for l in self._loci], dtype=np.uint32),
            minlen=self.minlen,
            support=self.support,
            unphased=self.unphased,
        )
