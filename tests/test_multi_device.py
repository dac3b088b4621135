"""The one-process multi-GPU entry (inq_genotype_repeats_devices, host/multi_device.cc; the reference's counterpart is the rayon
loop over loci, src/call.rs:103-145) as far as it can be covered without a GPU: the partition of the targets, one thread per part,
the scatter of the parts' rows and the ordered `.inq` text, through inq_host_devices_selftest (rows that name their target and
their part); the reader-pool share; and the bounded waits for a device context that never comes up (host/driver_internal.h
AsyncCtx - round 4 lost a GPU box to a context thread that aborted inside a profiler's signal handler)."""
import io
import os
import time

import numpy as np
import pytest

from inquistr_amd import call
from tests.test_host_frontend import _make_case


def _rows(text):
    lines = text.splitlines()
    assert lines[0] == "chromosome\tbegin\tend\tS_H1\tS_H2"
    return [ln.split("\t") for ln in lines[1:]]


@pytest.mark.parametrize("threads", [1, 4])
@pytest.mark.parametrize("n_parts", [1, 2, 3, 8, 64])
def test_parts_cover_every_target_once_in_the_reference_order(tmp_path, n_parts, threads):
    bam, bed, loci, _recs = _make_case(tmp_path, 51, n_loci=70)
    out = tmp_path / "self.inq"
    with open(out, "w") as f:
        cuts = call.devices_selftest(bam, None, bed, n_parts, f, threads=threads)
    rows = _rows(open(out).read())
    assert len(rows) == len(loci)
    # phase1 of a row = the position of its target in the BED: every target exactly once, coordinates intact
    seen = sorted(int(r[3]) for r in rows)
    assert seen == list(range(len(loci)))
    for r in rows:
        c, s, e, _t = loci[int(r[3])]
        assert (r[0], int(r[1]), int(r[2])) == (c, s, e)
    if threads == 1:  # BED order (src/call.rs:149-157)
        assert [int(r[3]) for r in rows] == list(range(len(loci)))
    else:  # (human chrom, start) order (src/call.rs:141); equal keys keep BED order
        keys = [(call.load().inq_host_human_compare(a[0].encode(), b[0].encode()), int(a[1]), int(b[1])) for a, b in zip(rows, rows[1:])]
        assert all(k[0] < 0 or (k[0] == 0 and k[1] <= k[2]) for k in keys)
    # the cuts are those of inq_run_partition, and a part is a contiguous stretch of the file-ordered targets
    run = call.Run(bam, None, bed, threads=threads)
    order, want_cuts = run.partition(n_parts)
    run.close()
    assert list(cuts) == list(want_cuts)
    part_of = {int(r[3]): int(r[4]) for r in rows}
    for p in range(n_parts):
        assert all(part_of[int(t)] == p for t in order[int(cuts[p]):int(cuts[p + 1])])
    assert int(cuts[0]) == 0 and int(cuts[-1]) == len(loci) and all(a <= b for a, b in zip(cuts, cuts[1:]))


def test_a_failing_part_ends_the_call_with_its_status(tmp_path):
    bam, bed, _loci, _recs = _make_case(tmp_path, 52, n_loci=20)
    with open(tmp_path / "x.inq", "w") as f:
        with pytest.raises(call.CallError) as ei:
            call.devices_selftest(bam, None, bed, 4, f, fail_part=2)
    assert ei.value.status == 101 and ei.value.message.startswith("part 2 of 4: injected failure")
    assert os.path.getsize(tmp_path / "x.inq") == 0  # nothing is written when a part fails (the reference panics before its output stage)


def test_device_list_is_validated(tmp_path):
    bam, bed, _loci, _recs = _make_case(tmp_path, 53, n_loci=5)
    with pytest.raises(call.CallError) as ei:
        call.genotype_repeats_devices(bam, None, bed, [], out=io.open(os.devnull, "w"))
    assert ei.value.status == 1
    with pytest.raises(call.CallError) as ei:
        call.genotype_repeats_devices(bam, None, bed, [0, -1], out=io.open(os.devnull, "w"))
    assert ei.value.status == 1


def test_reader_pool_takes_this_callers_share_of_the_granted_cores():
    L = call.load()
    granted = L.inq_host_granted_cpus()
    assert 1 <= granted <= len(os.sched_getaffinity(0))
    one = L.inq_host_span_io_threads(16, 1)
    assert one == min(16, max(2, granted), 32)
    for sharers in (2, 4, 8):
        got = L.inq_host_span_io_threads(16, sharers)
        assert got == min(16, max(2, granted // sharers))
        assert got * sharers <= max(granted, 2 * sharers)  # all sharers together stay within the grant (two threads each at least)
    # the default share comes from torch.distributed.run's environment, an explicit word overrides it
    os.environ["LOCAL_WORLD_SIZE"], os.environ["LOCAL_RANK"] = "4", "1"
    try:
        assert L.inq_host_span_io_threads(16, 0) == L.inq_host_span_io_threads(16, 4)
        L.inq_host_set_local_share(2, 0)
        assert L.inq_host_span_io_threads(16, 0) == L.inq_host_span_io_threads(16, 2)
    finally:
        L.inq_host_set_local_share(0, 0)
        del os.environ["LOCAL_WORLD_SIZE"], os.environ["LOCAL_RANK"]


@pytest.mark.skipif("tsan" in os.environ.get("LD_PRELOAD", ""), reason="starts child processes: a fork of the instrumented, threaded test process hangs in ThreadSanitizer")
def test_a_caller_short_of_cores_gets_blocking_waits_unless_its_user_said_otherwise():
    """The two cores that spin while a span inflates and crosses the link are the readers' when a caller's share of the granted cores
    is small (tools/few_cores.sh: + 7 ... 15 % span loop with the process confined to 2 - 4 CPUs): the host library then makes its
    contexts with "blocking_sync" - a process default set when the context thread is started - and leaves the option alone when the
    user has set it either way, or when the share is 8 cores or more.  Fresh processes: the default is the process's."""
    import subprocess
    import sys

    prog = """
import ctypes as C, sys
from inquistr_amd import call, hipcall
L, H = call.load(), hipcall.load()
sharers, user = int(sys.argv[1]), sys.argv[2]
if user != "-":
    assert H.inq_default_option(b"blocking_sync", int(user)) == 0
L.inq_host_set_local_share(sharers, 0)
v = C.c_int64(-1)
assert (H.inq_default_option_get(b"blocking_sync", C.byref(v)) == 0) == (user != "-")
S = call.Session(0)  # starts the context thread (there is no GPU here: the context fails later, the choice is made before)
rc = H.inq_default_option_get(b"blocking_sync", C.byref(v))
print(rc, v.value, L.inq_host_granted_cpus())
S.close()
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(sharers, user="-"):
        r = subprocess.run([sys.executable, "-c", prog, str(sharers), user], capture_output=True, text=True, cwd=root, timeout=120)
        assert r.returncode == 0, r.stderr[-1500:]
        rc, v, granted = r.stdout.split()[-3:]
        return int(rc), int(v), int(granted)

    rc, v, granted = run(64)  # a 64th of the grant: far below 8 cores
    assert granted // 64 < 8 and (rc, v) == (0, 1)
    assert run(64, "0")[:2] == (0, 0) and run(64, "1")[:2] == (0, 1)  # the user's word stays
    if granted >= 8:
        assert run(1)[0] != 0  # a caller with the whole grant: nothing is set, waits spin (the lower latency)


# ---- a device context that never comes up: every waiter is bounded, the call ends with exit status 1 ----
MODES = {1: "returns 'no device' at once", 2: "hangs in its first call, nothing published",
         3: "publishes a staging context, then hangs", 4: "publishes a staging context, then fails"}


@pytest.mark.timeout(60)
@pytest.mark.parametrize("mode", sorted(MODES))
@pytest.mark.parametrize("entry", ["call", "devices"])
def test_context_thread_that_dies_leads_to_an_error_exit_within_the_timeout(tmp_path, mode, entry):
    bam, bed, _loci, _recs = _make_case(tmp_path, 54, n_loci=25)
    L = call.load()
    timeout_ms = 1500
    L.inq_host_test_ctx_creator(mode, timeout_ms)
    try:
        t0 = time.perf_counter()
        with open(tmp_path / "o.inq", "w") as f:
            with pytest.raises(call.CallError) as ei:
                if entry == "call":
                    call.genotype_repeats(bam, None, bed, 5, 3, 2, False, "S", out=f, frontend="device")
                else:
                    call.genotype_repeats_devices(bam, None, bed, [0, 0], threads=2, sample_name="S", out=f, frontend="device")
        dt = time.perf_counter() - t0
    finally:
        L.inq_host_test_ctx_creator(0, -1)  # releases the threads that were left behind
    assert ei.value.status == 1, ei.value.message
    assert "cannot open HIP device" in ei.value.message
    if mode in (2, 3):
        assert "did not come up" in ei.value.message
        assert timeout_ms / 1e3 * 0.9 <= dt < timeout_ms / 1e3 + 4.0, dt
    else:
        assert dt < 4.0, dt
    assert os.path.getsize(tmp_path / "o.inq") == 0


@pytest.mark.timeout(60)
def test_host_sweep_front_end_is_bounded_too(tmp_path):
    bam, bed, _loci, _recs = _make_case(tmp_path, 55, n_loci=10)
    L = call.load()
    L.inq_host_test_ctx_creator(2, 1000)
    try:
        t0 = time.perf_counter()
        with pytest.raises(call.CallError) as ei:
            call.genotype_repeats(bam, None, bed, 5, 3, 1, False, "S", out=io.open(os.devnull, "w"), frontend="host")
        assert time.perf_counter() - t0 < 5.0
    finally:
        L.inq_host_test_ctx_creator(0, -1)
    assert ei.value.status == 1 and "did not come up" in ei.value.message
