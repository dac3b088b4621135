"""ctypes binding of the C ABI (include/inquistr_hip.h) — the only way Python reaches the
HIP kernels.  No CPU fallback: if the shared library is missing, or no gfx950 device is
visible, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from typing import Optional, Tuple

import numpy as np

from .batch import (
    INQ_ERR_ARG,
    INQ_ERR_AUX,
    INQ_ERR_LOCUS,
    INQ_ERR_BAM,
    INQ_ERR_INFLATE,
    INQ_ERR_NO_DEVICE,
    INQ_OK,
    Batch,
    InqBatchC,
    InqResultC,
    Result,
)

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libinquistr_hip.so")

# every symbol include/inquistr_hip.h declares
ABI_SYMBOLS = (
    "inq_ctx_create",
    "inq_ctx_destroy",
    "inq_call_batch",
    "inq_call_batch_device",
    "inq_ctx_status",
    "inq_ctx_timing_enable",
    "inq_ctx_timing_read",
    "inq_ctx_timing_reset",
    "inq_ctx_set_option",
    "inq_alloc_pinned",
    "inq_free_pinned",
    "inq_pin_host",
    "inq_unpin_host",
    "inq_strerror",
    "inq_backend_name",
    "inq_last_error",
    "inq_abi_version",
    "inq_ctx_numa_node",
    "inq_bgzf_inflate",
    "inq_call_span",
    "inq_span_stage",
    "inq_ctx_create_early",
    "inq_ctx_create_multi",
    "inq_default_option",
    "inq_default_option_get",
    "inq_ctx_alloc_retries",
    "inq_span_stage_begin",
    "inq_span_stage_wait",
    "inq_call_span_staged",
    "inq_call_span_deferred",
    "inq_call_flush",
    "inq_call_flush_device",
    "inq_dev_alloc_rows",
    "inq_dev_free_rows",
    "inq_dev_write_rows",
    "inq_dev_read_rows",
    "inq_call_deferred_loci",
    "inq_call_discard",
    "inq_span_fetch_batch",
    "inq_outlier_rows",
)


class BgzfBlockC(C.Structure):
    """inq_bgzf_block_t"""

    _fields_ = [("comp_off", C.c_uint64), ("comp_len", C.c_uint32), ("isize", C.c_uint32), ("out_off", C.c_uint64)]


BGZF_BLOCK_DTYPE = np.dtype([("comp_off", "<u8"), ("comp_len", "<u4"), ("isize", "<u4"), ("out_off", "<u8")])


ANCHOR_SEGMENT_END = 1 << 63


class SpanC(C.Structure):
    """inq_span_t"""

    _fields_ = [
        ("comp", C.c_void_p),
        ("comp_bytes", C.c_uint64),
        ("blocks", C.c_void_p),
        ("n_blocks", C.c_uint64),
        ("anchors", C.c_void_p),
        ("anchor_stop", C.c_void_p),
        ("n_anchors", C.c_uint64),
        ("locus_tid", C.c_void_p),
        ("locus_start", C.c_void_p),
        ("locus_end", C.c_void_p),
        ("n_loci", C.c_uint64),
        ("minlen", C.c_uint32),
        ("support", C.c_uint32),
        ("unphased", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class SpanStatsC(C.Structure):
    """inq_span_stats_t"""

    _fields_ = [
        ("n_records", C.c_uint64),
        ("n_reads", C.c_uint64),
        ("n_pairs", C.c_uint64),
        ("n_cigar_words", C.c_uint64),
        ("inflated_bytes", C.c_uint64),
        ("max_reads", C.c_uint32),
        ("front_status", C.c_uint32),
        ("first_bad_record", C.c_uint64),
        ("ms_upload", C.c_double),
        ("ms_inflate", C.c_double),
        ("ms_scan", C.c_double),
        ("ms_join", C.c_double),
        ("ms_call", C.c_double),
    ]


def scan_bgzf(data) -> np.ndarray:
    """Walks the BGZF headers of `data` (whole blocks): the block table inq_bgzf_inflate / inq_call_span take
    (payload offset and length, ISIZE, dense output offsets)."""
    mv = memoryview(data)
    out = []
    p, uo = 0, 0
    while p < len(mv):
        if p + 18 > len(mv) or mv[p] != 0x1F or mv[p + 1] != 0x8B or mv[p + 2] != 8 or not (mv[p + 3] & 4):
            raise ValueError(f"not a BGZF block at {p}")
        xlen = mv[p + 10] | (mv[p + 11] << 8)
        bsize = None
        q = p + 12
        while q + 4 <= p + 12 + xlen:
            slen = mv[q + 2] | (mv[q + 3] << 8)
            if mv[q] == 66 and mv[q + 1] == 67 and slen == 2:
                bsize = (mv[q + 4] | (mv[q + 5] << 8)) + 1
            q += 4 + slen
        if bsize is None or p + bsize > len(mv):
            raise ValueError(f"bad BGZF block at {p}")
        head = 12 + xlen
        isize = int.from_bytes(mv[p + bsize - 4 : p + bsize], "little")
        out.append((p + head, bsize - head - 8, isize, uo))
        uo += isize
        p += bsize
    return np.array(out, dtype=BGZF_BLOCK_DTYPE)


class InqError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"inquistr_hip error {code}: {msg}")
        self.code = code


_lib = None


def load(path: Optional[str] = None):
    """Loads libinquistr_hip.so (built by `make -C inquistr_amd/csrc` / __graft_entry__.build()).
    `path` loads another build of the library into its own handle (A/B timing of two builds in one
    process, tools/ab_options.py --libs); it is not cached."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # PyTorch wheels bundle their own libamdhip64.so.7; both it and /opt/rocm's carry the same SONAME, so
    # the first one loaded serves the whole process.  With ours first, torch later reports "No HIP GPUs
    # are available"; with torch's first, both work.  So when torch is installed (it provides device
    # memory and streams to bench.py and the tests) let it bring its runtime in before we load ours.
    if "torch" not in sys.modules and os.environ.get("INQ_SKIP_TORCH_PRELOAD") is None:
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    lib_path = path or LIB_PATH
    if not os.path.exists(lib_path):
        raise ImportError(
            f"{lib_path} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback."
        )
    L = C.CDLL(lib_path)
    vp = C.c_void_p
    L.inq_ctx_create.restype = C.c_int
    L.inq_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.inq_ctx_alloc_retries.restype = C.c_uint64
    L.inq_ctx_alloc_retries.argtypes = [vp]
    L.inq_default_option.restype = C.c_int
    L.inq_default_option.argtypes = [C.c_char_p, C.c_int64]
    L.inq_default_option_get.restype = C.c_int
    L.inq_default_option_get.argtypes = [C.c_char_p, C.POINTER(C.c_int64)]
    L.inq_ctx_create_multi.restype = C.c_int
    L.inq_ctx_create_multi.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    L.inq_ctx_destroy.restype = None
    L.inq_ctx_destroy.argtypes = [vp]
    L.inq_call_batch.restype = C.c_int
    L.inq_call_batch.argtypes = [vp, C.POINTER(InqBatchC), C.POINTER(InqResultC)]
    L.inq_call_batch_device.restype = C.c_int
    L.inq_call_batch_device.argtypes = [vp, C.POINTER(InqBatchC), C.POINTER(InqResultC), vp]
    L.inq_ctx_status.restype = C.c_int
    L.inq_ctx_status.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.inq_ctx_timing_enable.restype = C.c_int
    L.inq_ctx_timing_enable.argtypes = [vp, C.c_int]
    L.inq_ctx_timing_read.restype = C.c_int
    L.inq_ctx_timing_read.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.inq_ctx_timing_reset.restype = C.c_int
    L.inq_ctx_timing_reset.argtypes = [vp]
    L.inq_ctx_set_option.restype = C.c_int
    L.inq_ctx_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    L.inq_alloc_pinned.restype = C.c_int
    L.inq_alloc_pinned.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.inq_free_pinned.restype = None
    L.inq_free_pinned.argtypes = [vp]
    L.inq_pin_host.restype = C.c_int
    L.inq_pin_host.argtypes = [vp, C.c_size_t]
    L.inq_unpin_host.restype = None
    L.inq_unpin_host.argtypes = [vp]
    L.inq_strerror.restype = C.c_char_p
    L.inq_strerror.argtypes = [C.c_int]
    L.inq_backend_name.restype = C.c_char_p
    L.inq_backend_name.argtypes = [vp]
    L.inq_last_error.restype = C.c_char_p
    L.inq_last_error.argtypes = [vp]
    L.inq_abi_version.restype = C.c_int
    L.inq_abi_version.argtypes = []
    L.inq_ctx_numa_node.restype = C.c_int
    L.inq_ctx_numa_node.argtypes = [vp]
    L.inq_bgzf_inflate.restype = C.c_int
    L.inq_bgzf_inflate.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64, vp, C.c_uint64, vp]
    L.inq_call_span.restype = C.c_int
    L.inq_call_span.argtypes = [vp, C.POINTER(SpanC), C.POINTER(InqResultC), C.POINTER(SpanStatsC)]
    L.inq_span_stage.restype = C.c_int
    L.inq_span_stage.argtypes = [vp, C.POINTER(SpanC), C.c_int]
    L.inq_span_stage_begin.restype = C.c_int
    L.inq_span_stage_begin.argtypes = [vp, C.POINTER(SpanC), C.c_int]
    L.inq_span_stage_wait.restype = C.c_int
    L.inq_span_stage_wait.argtypes = [vp, C.c_int]
    L.inq_call_span_staged.restype = C.c_int
    L.inq_call_span_staged.argtypes = [vp, C.POINTER(SpanC), C.c_int, C.POINTER(InqResultC), C.POINTER(SpanStatsC)]
    L.inq_call_span_deferred.restype = C.c_int
    L.inq_call_span_deferred.argtypes = [vp, C.POINTER(SpanC), C.c_int, C.POINTER(SpanStatsC)]
    L.inq_call_flush.restype = C.c_int
    L.inq_call_flush.argtypes = [vp, C.POINTER(InqResultC), C.c_uint64, C.POINTER(C.c_double)]
    L.inq_call_deferred_loci.restype = C.c_uint64
    L.inq_call_deferred_loci.argtypes = [vp]
    L.inq_call_discard.restype = None
    L.inq_call_discard.argtypes = [vp]
    L.inq_outlier_rows.restype = C.c_int
    L.inq_outlier_rows.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint32, C.c_int, C.c_uint32, C.c_float, C.c_uint32, vp, vp]
    L.inq_span_fetch_batch.restype = C.c_int
    L.inq_span_fetch_batch.argtypes = [vp, vp, vp, vp, vp]
    if path is None:
        _lib = L
    return L


def strerror(code: int) -> str:
    return load().inq_strerror(code).decode()


class Context:
    """One HIP device context (inq_ctx_t).  Raises InqError(INQ_ERR_NO_DEVICE) without an MI355X."""

    def __init__(self, device_id: int = 0, lib=None, _handle=None):
        self._L = lib if lib is not None else load()
        self._h = C.c_void_p()
        if _handle is not None:  # made by Context.create_multi
            self._h = C.c_void_p(_handle)
            return
        rc = self._L.inq_ctx_create(device_id, C.byref(self._h))
        if rc != INQ_OK:
            self._h = C.c_void_p()
            raise InqError(rc, strerror(rc))

    @classmethod
    def create_multi(cls, device_ids, lib=None):
        """inq_ctx_create_multi: one context per entry of device_ids, made concurrently; all or nothing."""
        L = lib if lib is not None else load()
        n = len(device_ids)
        ids = (C.c_int * n)(*[int(d) for d in device_ids])
        hs = (C.c_void_p * n)()
        rc = L.inq_ctx_create_multi(ids, n, hs)
        if rc != INQ_OK:
            raise InqError(rc, strerror(rc))
        return [cls(lib=L, _handle=hs[i]) for i in range(n)]

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.inq_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def alloc_retries(self) -> int:
        return int(self._L.inq_ctx_alloc_retries(self._h))

    @property
    def backend(self) -> str:
        return self._L.inq_backend_name(self._h).decode()

    def _raise(self, rc: int):
        detail = self._L.inq_last_error(self._h).decode()
        raise InqError(rc, strerror(rc) + (f" [{detail}]" if detail and rc == -8 else ""))

    def call_batch(self, batch: Batch, debug: bool = False, check: bool = True) -> Tuple[int, Result]:
        """Host-buffer entry (inq_call_batch).  Returns (code, Result); raises on error if check."""
        res = Result.alloc(batch, debug=debug)
        bc, rc_ = batch.as_c(), res.as_c()
        rc = self._L.inq_call_batch(self._h, C.byref(bc), C.byref(rc_))
        res.n_tie_loci = int(rc_.n_tie_loci)
        if rc != INQ_OK and check:
            self._raise(rc)
        return rc, res

    def call_batch_device(self, bc: InqBatchC, rc_: InqResultC, stream: Optional[int] = None) -> None:
        """Device-resident entry: pointers in bc / rc_ are device pointers; enqueue only."""
        rc = self._L.inq_call_batch_device(self._h, C.byref(bc), C.byref(rc_), C.c_void_p(stream or 0))
        if rc != INQ_OK:
            self._raise(rc)

    def bgzf_inflate(self, comp, blocks: np.ndarray, check: bool = True):
        """inq_bgzf_inflate on host buffers: returns (code, inflated bytes as uint8 array, per-block status)."""
        comp = np.frombuffer(comp, dtype=np.uint8)
        blocks = np.ascontiguousarray(blocks, dtype=BGZF_BLOCK_DTYPE)
        n = len(blocks)
        total = int((blocks["out_off"] + blocks["isize"]).max()) if n else 0
        out = np.zeros(total, dtype=np.uint8)
        status = np.zeros(n, dtype=np.uint32)
        rc = self._L.inq_bgzf_inflate(self._h, comp.ctypes.data, comp.size, blocks.ctypes.data, n, out.ctypes.data, total,
                                      status.ctypes.data)
        if rc != INQ_OK and check:
            self._raise(rc)
        return rc, out, status

    def call_span(self, comp, blocks: np.ndarray, anchors: np.ndarray, anchor_stop: np.ndarray, locus_tid: np.ndarray,
                  locus_start: np.ndarray, locus_end: np.ndarray, minlen: int = 5, support: int = 3, unphased: bool = False,
                  check: bool = True, stage_slot: Optional[int] = None):
        """inq_call_span: returns (code, phase1, phase2, n_tie_loci, stats).  stage_slot: go through
        inq_span_stage + inq_call_span_staged with that slot instead."""
        comp = np.frombuffer(comp, dtype=np.uint8)
        blocks = np.ascontiguousarray(blocks, dtype=BGZF_BLOCK_DTYPE)
        anchors = np.ascontiguousarray(anchors, dtype=np.uint64)
        stops = np.ascontiguousarray(anchor_stop, dtype=np.uint64)
        lt = np.ascontiguousarray(locus_tid, dtype=np.int32)
        ls = np.ascontiguousarray(locus_start, dtype=np.uint32)
        le = np.ascontiguousarray(locus_end, dtype=np.uint32)
        assert len(stops) == len(anchors) and len(lt) == len(ls) == len(le)
        sp = SpanC(comp.ctypes.data, comp.size, blocks.ctypes.data, len(blocks), anchors.ctypes.data, stops.ctypes.data,
                   len(anchors), lt.ctypes.data, ls.ctypes.data, le.ctypes.data, len(ls), minlen, support, 1 if unphased else 0, 0)
        p1 = np.full(len(ls), np.nan)
        p2 = np.full(len(ls), np.nan)
        res = InqResultC(p1.ctypes.data, p2.ctypes.data, None, None, 0)
        stats = SpanStatsC()
        if stage_slot is None:
            rc = self._L.inq_call_span(self._h, C.byref(sp), C.byref(res), C.byref(stats))
        else:
            rc = self._L.inq_span_stage(self._h, C.byref(sp), stage_slot)
            if rc == INQ_OK:
                rc = self._L.inq_call_span_staged(self._h, C.byref(sp), stage_slot, C.byref(res), C.byref(stats))
        if rc != INQ_OK and check:
            self._raise(rc)
        return rc, p1, p2, int(res.n_tie_loci), stats

    def span_stage_begin(self, comp, blocks, anchors, anchor_stop, slot: int, check: bool = True) -> int:
        """inq_span_stage_begin: the upload (and the inflate behind it) of a span is enqueued into `slot`; `comp` must stay alive and
        unchanged until span_stage_wait(slot).  Block table and anchors are copied on the spot."""
        comp = np.frombuffer(comp, dtype=np.uint8)
        blocks = np.ascontiguousarray(blocks, dtype=BGZF_BLOCK_DTYPE)
        anchors = np.ascontiguousarray(anchors, dtype=np.uint64)
        stops = np.ascontiguousarray(anchor_stop, dtype=np.uint64)
        sp = SpanC(comp.ctypes.data, comp.size, blocks.ctypes.data, len(blocks), anchors.ctypes.data, stops.ctypes.data,
                   len(anchors), None, None, None, 0, 5, 3, 0, 0)
        rc = self._L.inq_span_stage_begin(self._h, C.byref(sp), slot)
        if rc != INQ_OK and check:
            self._raise(rc)
        return rc

    def span_stage_wait(self, slot: int, check: bool = True) -> int:
        rc = self._L.inq_span_stage_wait(self._h, slot)
        if rc != INQ_OK and check:
            self._raise(rc)
        return rc

    def call_span_deferred(self, comp, blocks, anchors, anchor_stop, locus_tid, locus_start, locus_end, minlen: int = 5, support: int = 3,
                           unphased: bool = False, check: bool = True, stage_slot: Optional[int] = None, prestaged: bool = False):
        """inq_call_span_deferred: appends the span's batch to the deferred one; returns (code, stats).  Rows: call_flush().
        prestaged: the span already sits in stage_slot (span_stage_begin + span_stage_wait): it is not staged again."""
        comp = np.frombuffer(comp, dtype=np.uint8)
        blocks = np.ascontiguousarray(blocks, dtype=BGZF_BLOCK_DTYPE)
        anchors = np.ascontiguousarray(anchors, dtype=np.uint64)
        stops = np.ascontiguousarray(anchor_stop, dtype=np.uint64)
        lt = np.ascontiguousarray(locus_tid, dtype=np.int32)
        ls = np.ascontiguousarray(locus_start, dtype=np.uint32)
        le = np.ascontiguousarray(locus_end, dtype=np.uint32)
        sp = SpanC(comp.ctypes.data, comp.size, blocks.ctypes.data, len(blocks), anchors.ctypes.data, stops.ctypes.data,
                   len(anchors), lt.ctypes.data, ls.ctypes.data, le.ctypes.data, len(ls), minlen, support, 1 if unphased else 0, 0)
        stats = SpanStatsC()
        rc = INQ_OK
        if stage_slot is not None and not prestaged:
            rc = self._L.inq_span_stage(self._h, C.byref(sp), stage_slot)
        if rc == INQ_OK:
            rc = self._L.inq_call_span_deferred(self._h, C.byref(sp), -1 if stage_slot is None else stage_slot, C.byref(stats))
        if rc != INQ_OK and check:
            self._raise(rc)
        return rc, stats

    def call_discard(self) -> None:
        self._L.inq_call_discard(self._h)

    @property
    def deferred_loci(self) -> int:
        return int(self._L.inq_call_deferred_loci(self._h))

    def call_flush(self, check: bool = True):
        """inq_call_flush: (code, phase1, phase2, n_tie_loci, ms_call) for every locus deferred since the last flush."""
        n = self.deferred_loci
        p1 = np.full(n, np.nan)
        p2 = np.full(n, np.nan)
        res = InqResultC(p1.ctypes.data, p2.ctypes.data, None, None, 0)
        ms = C.c_double(0.0)
        rc = self._L.inq_call_flush(self._h, C.byref(res), n, C.byref(ms))
        if rc != INQ_OK and check:
            self._raise(rc)
        return rc, p1, p2, int(res.n_tie_loci), float(ms.value)

    def span_fetch_batch(self, stats: "SpanStatsC", n_loci: int):
        """The batch the last call_span built on the device, as host arrays (cigar, reads, pair_read, locus_pair_off)."""
        from .batch import READ_DTYPE

        cigar = np.zeros(int(stats.n_cigar_words), dtype=np.uint32)
        reads = np.zeros(int(stats.n_reads), dtype=READ_DTYPE)
        pair_read = np.zeros(int(stats.n_pairs), dtype=np.uint32)
        off = np.zeros(n_loci + 1, dtype=np.uint64)
        rc = self._L.inq_span_fetch_batch(self._h, cigar.ctypes.data, reads.ctypes.data, pair_read.ctypes.data, off.ctypes.data)
        if rc != INQ_OK:
            self._raise(rc)
        return cigar, reads, pair_read, off

    def outlier_rows(self, values: np.ndarray, row_len: np.ndarray, method: str = "zscore", minsize: int = 10,
                     zscore_cutoff: float = 3.0, mincluster: int = 1, check: bool = True):
        """inq_outlier_rows: values f32 [n_rows, stride]; returns (code, flags u8 [n_rows, stride], keep u8 [n_rows])."""
        values = np.ascontiguousarray(values, dtype=np.float32)
        row_len = np.ascontiguousarray(row_len, dtype=np.uint32)
        n_rows, stride = values.shape if values.ndim == 2 else (len(row_len), 0)
        flags = np.zeros((n_rows, stride), dtype=np.uint8)
        keep = np.zeros(n_rows, dtype=np.uint8)
        rc = self._L.inq_outlier_rows(self._h, values.ctypes.data, row_len.ctypes.data, n_rows, stride,
                                      {"zscore": 0, "dbscan": 1}[method], minsize, zscore_cutoff, mincluster,
                                      flags.ctypes.data, keep.ctypes.data)
        if rc != INQ_OK and check:
            self._raise(rc)
        return rc, flags, keep

    def status(self) -> Tuple[int, int]:
        ties = C.c_uint64(0)
        rc = self._L.inq_ctx_status(self._h, C.byref(ties))
        return rc, int(ties.value)

    def timing_enable(self, on: bool = True):
        self._L.inq_ctx_timing_enable(self._h, 1 if on else 0)

    def timing_reset(self):
        rc = self._L.inq_ctx_timing_reset(self._h)
        if rc != INQ_OK:
            self._raise(rc)

    def timing_read(self, which: int = 0) -> Tuple[float, int]:
        ms, n = C.c_double(0), C.c_uint64(0)
        rc = self._L.inq_ctx_timing_read(self._h, which, C.byref(ms), C.byref(n))
        if rc != INQ_OK:
            self._raise(rc)
        return float(ms.value), int(n.value)

    def set_option(self, key: str, value: int):
        rc = self._L.inq_ctx_set_option(self._h, key.encode(), value)
        if rc != INQ_OK:
            self._raise(rc)
