"""Host-side mirror of the reference's `call::genotype_repeats` (src/call.rs:76-159).

Same argument list and meaning; the work is done by libinquistr_host.so (C++ BAM front end +
driver) which drives the HIP library.  The `.inq` text goes to `out` (default: stdout), a
non-zero status raises `CallError` carrying the exit code the reference process would have had
(1 for its explicit `exit(1)` paths, 101 for its panics).
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from typing import Iterator, Optional, Tuple

import numpy as np

from .batch import READ_DTYPE, Batch, InqBatchC

_PKG = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.environ.get("INQ_HOST_LIB") or os.path.join(_PKG, "lib", "libinquistr_host.so")  # INQ_HOST_LIB: sanitizer build in CI
CLI_PATH = os.path.join(_PKG, "lib", "inquistr")

HOST_ABI_SYMBOLS = (
    "inq_genotype_repeats",
    "inq_genotype_repeats_rows",
    "inq_genotype_repeats_devices",
    "inq_host_set_local_share",
    "inq_host_granted_cpus",
    "inq_host_ctx_option",
    "inq_host_last_call_stats",
    "inq_host_span_io_threads",
    "inq_host_devices_selftest",
    "inq_host_test_ctx_creator",
    "inq_host_partition",
    "inq_run_open",
    "inq_run_n_targets",
    "inq_run_sample",
    "inq_run_target",
    "inq_run_partition",
    "inq_run_rows",
    "inq_run_rows_device",
    "inq_run_write_inq",
    "inq_run_close",
    "inq_session_open",
    "inq_session_call",
    "inq_session_call_many",
    "inq_session_close",
    "inq_session_stage",
    "inq_session_run",
    "inq_session_discard",
    "inq_session_run_open",
    "inq_combine",
    "inq_frontend_open",
    "inq_frontend_n_targets",
    "inq_frontend_target",
    "inq_frontend_sample",
    "inq_frontend_next",
    "inq_frontend_set_batch_words",
    "inq_frontend_close",
    "inq_host_format_f64",
    "inq_host_format_row",
    "inq_host_format_header",
    "inq_host_sample_name",
    "inq_host_human_compare",
    "inq_host_parse_region",
    "inq_host_bai_stats",
    "inq_host_span_bytes_read",
    "inq_host_iopool_selftest",
    "inq_host_bai_file_offset",
    "inq_host_bai_scan_start",
    "inq_host_plan_spans",
    "inq_host_bam_tid",
    "inq_spans_open",
    "inq_spans_n_targets",
    "inq_spans_next",
    "inq_spans_close",
    "inq_outlier",
)


class CallArgsC(C.Structure):
    _fields_ = [
        ("bam", C.c_char_p),
        ("region", C.c_char_p),
        ("region_file", C.c_char_p),
        ("minlen", C.c_uint32),
        ("support", C.c_uint64),
        ("threads", C.c_uint64),
        ("unphased", C.c_int32),
        ("sample_name", C.c_char_p),
        ("reference", C.c_char_p),
        ("device", C.c_int32),
        ("reserved", C.c_int32),
    ]


class PartStatsC(C.Structure):
    """inq_part_stats_t: what one device part of inq_genotype_repeats_devices reports"""
    _fields_ = [
        ("device", C.c_int32),
        ("status", C.c_int32),
        ("loci", C.c_uint64),
        ("spans", C.c_uint64),
        ("bam_bytes_read", C.c_uint64),
        ("rows_s", C.c_double),
        ("span_loop_s", C.c_double),
        ("wait_loader_s", C.c_double),
        ("device_calls_s", C.c_double),
        ("front", C.c_int32),
        ("io_threads", C.c_int32),
    ]


class OutlierArgsC(C.Structure):
    _fields_ = [
        ("combined", C.c_char_p),
        ("minsize", C.c_uint32),
        ("zscore", C.c_float),
        ("method", C.c_int32),
        ("sample", C.c_char_p),
        ("subset_file", C.c_char_p),
        ("device", C.c_int32),
        ("reserved", C.c_int32),
    ]


class CallError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"inquistr call failed (exit status {status}): {message}")
        self.status, self.message = status, message


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise ImportError(f"{HOST_LIB_PATH} not found: run __graft_entry__.build()")
        from . import hipcall

        hipcall.load()  # HIP runtime load order (see hipcall.load) + the library this one links against
        L = C.CDLL(HOST_LIB_PATH)
        vp = C.c_void_p
        L.inq_genotype_repeats.restype = C.c_int
        L.inq_genotype_repeats.argtypes = [C.POINTER(CallArgsC), C.c_int, C.c_char_p, C.c_size_t]
        L.inq_genotype_repeats_rows.restype = C.c_int
        L.inq_genotype_repeats_rows.argtypes = [C.POINTER(CallArgsC), C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
        L.inq_genotype_repeats_devices.restype = C.c_int
        L.inq_genotype_repeats_devices.argtypes = [C.POINTER(CallArgsC), C.POINTER(C.c_int32), C.c_size_t, C.c_int, C.POINTER(PartStatsC), C.c_char_p, C.c_size_t]
        L.inq_host_set_local_share.restype = None
        L.inq_host_set_local_share.argtypes = [C.c_int, C.c_int]
        L.inq_host_last_call_stats.restype = None
        L.inq_host_last_call_stats.argtypes = [C.POINTER(PartStatsC)]
        L.inq_host_ctx_option.restype = C.c_int
        L.inq_host_ctx_option.argtypes = [C.c_char_p, C.c_int64]
        L.inq_host_granted_cpus.restype = C.c_int
        L.inq_host_granted_cpus.argtypes = []
        L.inq_host_span_io_threads.restype = C.c_int
        L.inq_host_span_io_threads.argtypes = [C.c_uint64, C.c_int]
        L.inq_host_devices_selftest.restype = C.c_int
        L.inq_host_devices_selftest.argtypes = [C.POINTER(CallArgsC), C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_char_p, C.c_size_t]
        L.inq_host_test_ctx_creator.restype = None
        L.inq_host_test_ctx_creator.argtypes = [C.c_int, C.c_long]
        L.inq_host_partition.restype = C.c_int
        L.inq_host_partition.argtypes = [C.POINTER(CallArgsC), C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]
        L.inq_run_open.restype = C.c_int
        L.inq_run_open.argtypes = [C.POINTER(CallArgsC), C.POINTER(vp), C.c_char_p, C.c_size_t]
        L.inq_run_n_targets.restype = C.c_uint64
        L.inq_run_n_targets.argtypes = [vp]
        L.inq_run_sample.restype = C.c_char_p
        L.inq_run_sample.argtypes = [vp]
        L.inq_run_target.restype = C.c_int
        L.inq_run_target.argtypes = [vp, C.c_uint64, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.inq_run_partition.restype = C.c_int
        L.inq_run_partition.argtypes = [vp, C.c_uint64, C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
        L.inq_run_rows.restype = C.c_int
        L.inq_run_rows.argtypes = [vp, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
        L.inq_run_rows_device.restype = C.c_int
        L.inq_run_rows_device.argtypes = [vp, C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(vp), C.POINTER(vp), C.c_char_p, C.c_size_t]
        L.inq_run_write_inq.restype = C.c_int
        L.inq_run_write_inq.argtypes = [vp, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_char_p, C.c_size_t]
        L.inq_run_close.restype = None
        L.inq_run_close.argtypes = [vp]
        L.inq_session_open.restype = C.c_int
        L.inq_session_open.argtypes = [C.c_int32, C.POINTER(vp)]
        L.inq_session_call.restype = C.c_int
        L.inq_session_call.argtypes = [vp, C.POINTER(CallArgsC), C.c_int, C.c_char_p, C.c_size_t]
        L.inq_session_call_many.restype = C.c_int
        L.inq_session_call_many.argtypes = [vp, C.POINTER(CallArgsC), C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_size_t]
        L.inq_session_close.restype = None
        L.inq_session_close.argtypes = [vp]
        L.inq_session_run_open.restype = C.c_int
        L.inq_session_run_open.argtypes = [vp, C.POINTER(CallArgsC), C.POINTER(vp), C.c_char_p, C.c_size_t]
        L.inq_combine.restype = C.c_int
        L.inq_combine.argtypes = [C.POINTER(C.c_char_p), C.c_size_t, C.c_int, C.c_char_p, C.c_size_t]
        L.inq_frontend_open.restype = C.c_int
        L.inq_frontend_open.argtypes = [C.POINTER(CallArgsC), C.POINTER(vp), C.c_char_p, C.c_size_t]
        L.inq_frontend_n_targets.restype = C.c_uint64
        L.inq_frontend_n_targets.argtypes = [vp]
        L.inq_frontend_target.restype = C.c_int
        L.inq_frontend_target.argtypes = [vp, C.c_uint64, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.inq_frontend_sample.restype = C.c_char_p
        L.inq_frontend_sample.argtypes = [vp]
        L.inq_frontend_next.restype = C.c_int
        L.inq_frontend_next.argtypes = [vp, C.POINTER(InqBatchC), C.POINTER(C.POINTER(C.c_uint32)), C.c_char_p, C.c_size_t]
        L.inq_frontend_set_batch_words.restype = None
        L.inq_frontend_set_batch_words.argtypes = [vp, C.c_uint64]
        L.inq_frontend_close.restype = None
        L.inq_frontend_close.argtypes = [vp]
        L.inq_host_format_f64.restype = C.c_size_t
        L.inq_host_format_f64.argtypes = [C.c_double, C.c_char_p, C.c_size_t]
        L.inq_host_format_row.restype = C.c_size_t
        L.inq_host_format_row.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_char_p, C.c_size_t]
        L.inq_host_format_header.restype = C.c_size_t
        L.inq_host_format_header.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.inq_host_sample_name.restype = C.c_size_t
        L.inq_host_sample_name.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.inq_host_human_compare.restype = C.c_int
        L.inq_host_human_compare.argtypes = [C.c_char_p, C.c_char_p]
        L.inq_host_parse_region.restype = C.c_int
        L.inq_host_parse_region.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.inq_host_bai_file_offset.restype = C.c_uint64
        L.inq_host_bai_file_offset.argtypes = [C.c_char_p, C.c_int32, C.c_int64]
        L.inq_host_bai_scan_start.restype = C.c_uint64
        L.inq_host_bai_scan_start.argtypes = [C.c_char_p, C.c_int32, C.c_int64]
        L.inq_host_plan_spans.restype = C.c_int
        L.inq_host_plan_spans.argtypes = [C.POINTER(CallArgsC), C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                          C.c_void_p, C.c_uint64, C.c_char_p, C.c_size_t]
        L.inq_host_bam_tid.restype = C.c_int
        L.inq_host_bam_tid.argtypes = [C.c_char_p, C.c_char_p]
        L.inq_host_iopool_selftest.restype = C.c_uint64
        L.inq_host_iopool_selftest.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64]
        L.inq_host_span_bytes_read.restype = C.c_uint64
        L.inq_host_span_bytes_read.argtypes = []
        L.inq_host_bai_stats.restype = C.c_int
        L.inq_host_bai_stats.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        from .hipcall import SpanC

        L.inq_spans_open.restype = C.c_int
        L.inq_spans_open.argtypes = [C.POINTER(CallArgsC), C.c_uint64, C.POINTER(vp), C.c_char_p, C.c_size_t]
        L.inq_spans_n_targets.restype = C.c_uint64
        L.inq_spans_n_targets.argtypes = [vp]
        L.inq_spans_next.restype = C.c_int
        L.inq_spans_next.argtypes = [vp, C.POINTER(SpanC), C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]
        L.inq_spans_close.restype = None
        L.inq_spans_close.argtypes = [vp]
        L.inq_outlier.restype = C.c_int
        L.inq_outlier.argtypes = [C.POINTER(OutlierArgsC), C.c_int, C.c_char_p, C.c_size_t]
        _lib = L
    return _lib


FRONTENDS = {None: 0, "host": 1, "device": 2}


def _args(bamp, region, region_file, minlen, support, threads, unphased, sample_name, reference, device=0,
          frontend=None) -> CallArgsC:
    a = CallArgsC()
    a.bam = os.fspath(bamp).encode()
    a.region = region.encode() if region is not None else None
    a.region_file = os.fspath(region_file).encode() if region_file is not None else None
    a.minlen, a.support, a.threads = int(minlen), int(support), int(threads)
    a.unphased = 1 if unphased else 0
    a.sample_name = sample_name.encode() if sample_name is not None else None
    a.reference = reference.encode() if reference is not None else None
    a.device = device
    a.reserved = FRONTENDS[frontend]
    return a


def genotype_repeats(bamp: str, region: Optional[str], region_file: Optional[str], minlen: int = 5, support: int = 3,
                     threads: int = 1, unphased: bool = False, sample_name: Optional[str] = None,
                     reference: Optional[str] = None, out=None, device: int = 0, frontend: Optional[str] = None) -> None:
    """src/call.rs:76-86: same parameters, same output; raises CallError instead of exiting.
    frontend: "host" (BGZF inflate + record decode on CPU threads), "device" (inq_call_span: both on the GPU)
    or None (env INQ_FRONTEND, else the library default)."""
    L = load()
    a = _args(bamp, region, region_file, minlen, support, threads, unphased, sample_name, reference, device, frontend)
    err = C.create_string_buffer(2048)
    out = sys.stdout if out is None else out
    out.flush()
    fd = out.fileno()
    rc = L.inq_genotype_repeats(C.byref(a), fd, err, len(err))
    if rc != 0:
        raise CallError(rc, err.value.decode(errors="replace"))


def genotype_repeats_devices(bamp: str, region: Optional[str], region_file: Optional[str], devices, minlen: int = 5, support: int = 3,
                             threads: int = 1, unphased: bool = False, sample_name: Optional[str] = None, out=None,
                             frontend: Optional[str] = None):
    """inq_genotype_repeats_devices: the same command on several HIP devices from this one process (one thread + one device
    context per entry of `devices`; an ordinal may repeat).  Returns the per-part statistics as a list of dicts."""
    L = load()
    a = _args(bamp, region, region_file, minlen, support, threads, unphased, sample_name, None, 0, frontend)
    ids = (C.c_int32 * len(devices))(*[int(d) for d in devices])
    st = (PartStatsC * len(devices))()
    err = C.create_string_buffer(2048)
    out = sys.stdout if out is None else out
    out.flush()
    rc = L.inq_genotype_repeats_devices(C.byref(a), ids, len(devices), out.fileno(), st, err, len(err))
    if rc != 0:
        raise CallError(rc, err.value.decode(errors="replace"))
    return [{f: getattr(x, f) for f, _ in PartStatsC._fields_} for x in st]


def last_call_stats() -> dict:
    """inq_host_last_call_stats: spans, BAM bytes, span-loop / loader / device seconds, front end, reader threads of the last call."""
    st = PartStatsC()
    load().inq_host_last_call_stats(C.byref(st))
    return {f: getattr(st, f) for f in ("spans", "bam_bytes_read", "span_loop_s", "wait_loader_s", "device_calls_s", "front", "io_threads")}


def devices_selftest(bamp: str, region: Optional[str], region_file: Optional[str], n_parts: int, out, threads: int = 1, fail_part: int = -1):
    """inq_host_devices_selftest (no GPU): partition, one thread per part, scatter and ordered text of the multi-device entry with
    rows phase1 = position of the target in the list, phase2 = the part that called it.  Returns the cut points."""
    L = load()
    a = _args(bamp, region, region_file, 5, 3, threads, False, "S", None)
    cuts = np.zeros(n_parts + 1, dtype=np.uint64)
    err = C.create_string_buffer(2048)
    out.flush()
    rc = L.inq_host_devices_selftest(C.byref(a), n_parts, fail_part, out.fileno(), cuts.ctypes.data, err, len(err))
    if rc != 0:
        raise CallError(rc, err.value.decode(errors="replace"))
    return cuts.astype(np.int64)


def genotype_repeats_rows(bamp: str, region: Optional[str], region_file: Optional[str], target_index, minlen: int = 5, support: int = 3,
                          threads: int = 1, unphased: bool = False, device: int = 0, frontend: Optional[str] = None):
    """inq_genotype_repeats_rows: the rows (phase1, phase2 as f64 arrays) of the targets `target_index` points at."""
    L = load()
    idx = np.ascontiguousarray(target_index, dtype=np.uint32)
    a = _args(bamp, region, region_file, minlen, support, threads, unphased, None, None, device, frontend)
    p1 = np.full(len(idx), np.nan)
    p2 = np.full(len(idx), np.nan)
    err = C.create_string_buffer(2048)
    rc = L.inq_genotype_repeats_rows(C.byref(a), idx.ctypes.data, len(idx), p1.ctypes.data, p2.ctypes.data, err, len(err))
    if rc != 0:
        raise CallError(rc, err.value.decode(errors="replace"))
    return p1, p2


def partition(bamp: str, region: Optional[str], region_file: Optional[str], world: int):
    """inq_host_partition: (order, cuts) - the targets in file order and world + 1 cut points balanced by BAM bytes."""
    L = load()
    a = _args(bamp, region, region_file, 5, 3, 1, False, None, None)
    err = C.create_string_buffer(2048)
    n = C.c_uint64(0)
    cuts = np.zeros(world + 1, dtype=np.uint64)
    cap = 1 << 16
    while True:
        order = np.zeros(cap, dtype=np.uint32)
        rc = L.inq_host_partition(C.byref(a), world, order.ctypes.data, cap, cuts.ctypes.data, C.byref(n), err, len(err))
        if rc != 0 and n.value > cap:
            cap = int(n.value)
            continue
        if rc != 0:
            raise CallError(rc, err.value.decode(errors="replace"))
        return order[: n.value].copy(), cuts.astype(np.int64)


class Run:
    """inq_run_*: the BAM header, its index and the targets opened once (get_targets + get_bam_reader, src/call.rs:146-147,
    182-202); serves the work split, this process's rows and the ordered `.inq` output of a multi-process run.
    session: a Session whose device context, span buffers and BED cache the run's rows calls use (inq_session_run_open) - what a
    resident rank passes file after file; the session must stay open while the run is."""

    def __init__(self, bamp, region=None, region_file=None, minlen=5, support=3, threads=1, unphased=False, sample_name=None,
                 device: int = 0, frontend: Optional[str] = None, session: Optional["Session"] = None):
        self._L = load()
        self._h = C.c_void_p()
        self._session = session  # the session closes the runs still open on it before it goes (Session.close)
        self._args = _args(bamp, region, region_file, minlen, support, threads, unphased, sample_name, None, device, frontend)
        err = C.create_string_buffer(2048)
        if session is not None:
            rc = self._L.inq_session_run_open(session._h, C.byref(self._args), C.byref(self._h), err, len(err))
        else:
            rc = self._L.inq_run_open(C.byref(self._args), C.byref(self._h), err, len(err))
        if rc != 0:
            self._h = C.c_void_p()
            raise CallError(rc, err.value.decode(errors="replace"))
        if session is not None:
            session._runs.add(self)

    @property
    def n_targets(self) -> int:
        return int(self._L.inq_run_n_targets(self._h))

    @property
    def sample(self) -> str:
        return self._L.inq_run_sample(self._h).decode()

    def partition(self, world: int):
        n = self.n_targets
        order = np.zeros(max(n, 1), dtype=np.uint32)
        cuts = np.zeros(world + 1, dtype=np.uint64)
        err = C.create_string_buffer(2048)
        rc = self._L.inq_run_partition(self._h, world, order.ctypes.data, cuts.ctypes.data, err, len(err))
        if rc != 0:
            raise CallError(rc, err.value.decode(errors="replace"))
        return order[:n].copy(), cuts.astype(np.int64)

    def rows(self, target_index):
        idx = np.ascontiguousarray(target_index, dtype=np.uint32)
        p1 = np.full(len(idx), np.nan)
        p2 = np.full(len(idx), np.nan)
        err = C.create_string_buffer(2048)
        rc = self._L.inq_run_rows(self._h, idx.ctypes.data, len(idx), p1.ctypes.data, p2.ctypes.data, err, len(err))
        if rc != 0:
            raise CallError(rc, err.value.decode(errors="replace"))
        return p1, p2

    def rows_device(self, target_index, width: int):
        """inq_run_rows_device: the rows left in device memory - two device addresses of `width` f64 each (row k = target
        target_index[k], NaN behind the last), owned by the run and valid until its next call or close()."""
        idx = np.ascontiguousarray(target_index, dtype=np.uint32)
        d1, d2 = C.c_void_p(), C.c_void_p()
        err = C.create_string_buffer(2048)
        rc = self._L.inq_run_rows_device(self._h, idx.ctypes.data, len(idx), int(width), C.byref(d1), C.byref(d2), err, len(err))
        if rc != 0:
            raise CallError(rc, err.value.decode(errors="replace"))
        return int(d1.value), int(d2.value)

    def write_inq(self, phase1, phase2, out=None) -> None:
        """The output stage (src/call.rs:137-157) on rows in target-list order: one C call, whatever the row count."""
        p1 = np.ascontiguousarray(phase1, dtype=np.float64)
        p2 = np.ascontiguousarray(phase2, dtype=np.float64)
        out = sys.stdout if out is None else out
        out.flush()
        err = C.create_string_buffer(2048)
        rc = self._L.inq_run_write_inq(self._h, p1.ctypes.data, p2.ctypes.data, len(p1), out.fileno(), err, len(err))
        if rc != 0:
            raise CallError(rc, err.value.decode(errors="replace"))

    def close(self):
        if self._h and self._h.value:
            self._L.inq_run_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Session:
    """inq_session_*: many BAMs on one device context (the HIP runtime starts once; file k + 1 is staged while file k is called)."""

    def __init__(self, device: int = 0):
        import weakref

        self._L = load()
        self._h = C.c_void_p()
        self._runs = weakref.WeakSet()  # runs opened on this session (Run(session=...)): closed with it, never after it
        rc = self._L.inq_session_open(device, C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            raise CallError(rc, "cannot open a session")

    def call(self, bamp, region=None, region_file=None, minlen=5, support=3, threads=1, unphased=False, sample_name=None, out=None,
             frontend: Optional[str] = None) -> None:
        a = _args(bamp, region, region_file, minlen, support, threads, unphased, sample_name, None, 0, frontend)
        err = C.create_string_buffer(2048)
        out = sys.stdout if out is None else out
        out.flush()
        rc = self._L.inq_session_call(self._h, C.byref(a), out.fileno(), err, len(err))
        if rc != 0:
            raise CallError(rc, err.value.decode(errors="replace"))

    def call_many(self, bams, outs, region=None, region_file=None, minlen=5, support=3, threads=1, unphased=False, sample_names=None,
                  frontend: Optional[str] = None):
        """Runs the files in order; returns the list of exit statuses (no exception for failing files)."""
        n = len(bams)
        arr = (CallArgsC * n)()
        keep = []
        for k, b in enumerate(bams):
            a = _args(b, region, region_file, minlen, support, threads, unphased, sample_names[k] if sample_names else None, None, 0, frontend)
            keep.append(a)
            arr[k] = a
        for o in outs:
            o.flush()
        fds = (C.c_int * n)(*[o.fileno() for o in outs])
        st = (C.c_int * n)()
        err = C.create_string_buffer(2048)
        self._L.inq_session_call_many(self._h, arr, n, fds, st, err, len(err))
        self.last_message = err.value.decode(errors="replace")
        return list(st)

    def close(self):
        if self._h and self._h.value:
            for r in list(self._runs):  # a run uses the session's context to its end (its device rows are freed on it)
                r.close()
            self._L.inq_session_close(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def plan_spans(bamp, region=None, region_file=None, max_comp_bytes: int = 0, n_targets_cap: int = 1 << 20):
    """inq_host_plan_spans: (segments as an array of (vo_begin, vo_limit, span), per-target span number) - the plan alone."""
    L = load()
    a = _args(bamp, region, region_file, 5, 3, 1, False, None, None)
    err = C.create_string_buffer(2048)
    n = C.c_uint64(0)
    cap = 1024
    tspan = np.zeros(n_targets_cap, dtype=np.uint32)
    while True:
        vb, vl, sp = np.zeros(cap, dtype=np.uint64), np.zeros(cap, dtype=np.uint64), np.zeros(cap, dtype=np.uint32)
        rc = L.inq_host_plan_spans(C.byref(a), max_comp_bytes, vb.ctypes.data, vl.ctypes.data, sp.ctypes.data, cap, C.byref(n), tspan.ctypes.data,
                                   len(tspan), err, len(err))
        if rc != 0:
            raise CallError(rc, err.value.decode(errors="replace"))
        if n.value > cap:
            cap = int(n.value)
            continue
        k = int(n.value)
        return [(int(vb[i]), int(vl[i]), int(sp[i])) for i in range(k)], tspan


def combine(calls, out=None) -> None:
    """src/combine.rs:27-59: paste the H1/H2 columns of several .inq files next to the first one's rows."""
    L = load()
    out = sys.stdout if out is None else out
    out.flush()
    arr = (C.c_char_p * len(calls))(*[os.fspath(c).encode() for c in calls])
    err = C.create_string_buffer(2048)
    rc = L.inq_combine(arr, len(calls), out.fileno(), err, len(err))
    if rc != 0:
        raise CallError(rc, err.value.decode(errors="replace"))


def outlier(combined, minsize: int = 10, zscore: float = 3.0, method: str = "zscore", sample: Optional[str] = None,
            subset: Optional[str] = None, out=None, device: int = 0) -> None:
    """src/outlier.rs:33 `outlier(combined, minsize, zscore_cutoff, method, subset)` with the CLI's -s / -S (src/main.rs:
    213-227): loci of a combined .inq with outlying samples.  Raises CallError (101 where the reference panics)."""
    L = load()
    a = OutlierArgsC()
    a.combined = os.fspath(combined).encode()
    a.minsize, a.zscore = int(minsize), float(zscore)
    a.method = {"zscore": 0, "dbscan": 1}[method]
    a.sample = sample.encode() if sample is not None else None
    a.subset_file = os.fspath(subset).encode() if subset is not None else None
    a.device = device
    err = C.create_string_buffer(2048)
    out = sys.stdout if out is None else out
    out.flush()
    rc = L.inq_outlier(C.byref(a), out.fileno(), err, len(err))
    if rc != 0:
        raise CallError(rc, err.value.decode(errors="replace"))


class FrontEnd:
    """BAM + targets -> batches (the fetch()/rc_records() stage), no GPU involved."""

    def __init__(self, bamp, region=None, region_file=None, minlen=5, support=3, threads=1, unphased=False,
                 sample_name=None, max_batch_words: int = 0):
        self._L = load()
        self._h = C.c_void_p()
        self._args = _args(bamp, region, region_file, minlen, support, threads, unphased, sample_name, None)
        err = C.create_string_buffer(2048)
        rc = self._L.inq_frontend_open(C.byref(self._args), C.byref(self._h), err, len(err))
        if rc != 0:
            self._h = C.c_void_p()
            raise CallError(rc, err.value.decode(errors="replace"))
        if max_batch_words:
            self._L.inq_frontend_set_batch_words(self._h, max_batch_words)

    @property
    def sample(self) -> str:
        return self._L.inq_frontend_sample(self._h).decode()

    def targets(self):
        out = []
        for i in range(self._L.inq_frontend_n_targets(self._h)):
            c, s, e = C.c_char_p(), C.c_uint32(), C.c_uint32()
            self._L.inq_frontend_target(self._h, i, C.byref(c), C.byref(s), C.byref(e))
            out.append((c.value.decode(), s.value, e.value))
        return out

    def batches(self) -> Iterator[Tuple[Batch, np.ndarray]]:
        """Yields (Batch copy, locus_index) until the BAM is exhausted."""
        while True:
            bc = InqBatchC()
            idx = C.POINTER(C.c_uint32)()
            err = C.create_string_buffer(2048)
            rc = self._L.inq_frontend_next(self._h, C.byref(bc), C.byref(idx), err, len(err))
            if rc < 0:
                raise CallError(-rc, err.value.decode(errors="replace"))
            if rc == 0:
                return

            def arr(ptr, n, dt):
                if not n:
                    return np.zeros(0, dtype=dt)
                return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n * np.dtype(dt).itemsize,)).view(dt).copy()

            b = Batch(
                cigar=arr(bc.cigar, bc.n_cigar_words, np.uint32),
                reads=arr(bc.reads, bc.n_reads, READ_DTYPE),
                pair_read=arr(bc.pair_read, bc.n_pairs, np.uint32),
                locus_pair_off=arr(bc.locus_pair_off, bc.n_loci + 1, np.uint64),
                locus_start=arr(bc.locus_start, bc.n_loci, np.uint32),
                locus_end=arr(bc.locus_end, bc.n_loci, np.uint32),
                minlen=bc.minlen, support=bc.support, unphased=bool(bc.unphased),
            )
            yield b, np.ctypeslib.as_array(idx, shape=(max(int(bc.n_loci), 1),))[: int(bc.n_loci)].copy()

    def close(self):
        if self._h and self._h.value:
            self._L.inq_frontend_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Spans:
    """Host half of the device front end (inq_spans_*): yields, per span, everything inq_call_span takes.
    No GPU involved."""

    def __init__(self, bamp, region=None, region_file=None, minlen=5, support=3, threads=1, unphased=False,
                 max_comp_bytes: int = 0):
        self._L = load()
        self._h = C.c_void_p()
        self._args = _args(bamp, region, region_file, minlen, support, threads, unphased, None, None)
        err = C.create_string_buffer(2048)
        rc = self._L.inq_spans_open(C.byref(self._args), max_comp_bytes, C.byref(self._h), err, len(err))
        if rc != 0:
            self._h = C.c_void_p()
            raise CallError(rc, err.value.decode(errors="replace"))

    @property
    def n_targets(self) -> int:
        return int(self._L.inq_spans_n_targets(self._h))

    def spans(self):
        """Yields dicts of numpy copies: comp (u8), blocks, anchors / anchor_stop (u64), locus_tid/start/end,
        locus_index, file_begin."""
        from .hipcall import BGZF_BLOCK_DTYPE, SpanC

        while True:
            sp = SpanC()
            idx = C.POINTER(C.c_uint32)()
            fb = C.c_uint64(0)
            err = C.create_string_buffer(2048)
            rc = self._L.inq_spans_next(self._h, C.byref(sp), C.byref(idx), C.byref(fb), err, len(err))
            if rc < 0:
                raise CallError(-rc, err.value.decode(errors="replace"))
            if rc == 0:
                return

            def arr(ptr, n, dt):
                if not n:
                    return np.zeros(0, dtype=dt)
                return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n * np.dtype(dt).itemsize,)).view(dt).copy()

            n = int(sp.n_loci)
            yield dict(
                comp=arr(sp.comp, sp.comp_bytes, np.uint8), blocks=arr(sp.blocks, sp.n_blocks, BGZF_BLOCK_DTYPE),
                anchors=arr(sp.anchors, sp.n_anchors, np.uint64), anchor_stop=arr(sp.anchor_stop, sp.n_anchors, np.uint64),
                locus_tid=arr(sp.locus_tid, n, np.int32), locus_start=arr(sp.locus_start, n, np.uint32), locus_end=arr(sp.locus_end, n, np.uint32),
                locus_index=np.ctypeslib.as_array(idx, shape=(max(n, 1),))[:n].copy(), file_begin=int(fb.value),
                minlen=int(sp.minlen), support=int(sp.support), unphased=bool(sp.unphased),
            )

    def close(self):
        if self._h and self._h.value:
            self._L.inq_spans_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
