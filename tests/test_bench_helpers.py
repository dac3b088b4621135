"""bench.py's host-side helpers that can be checked without a GPU: the writer of the large BAM that works beside the earlier blocks
(BackgroundGen) must really stand still while it is stopped - every timed region of the default line relies on that - and finish the
file once it is let go; the core count the CPU legs and reader pools are sized from honours a cgroup quota."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _state(pid):
    with open(f"/proc/{pid}/stat") as f:
        return f.read().rsplit(")", 1)[1].split()[0]


def test_background_writer_stops_when_paused_and_finishes_when_released(tmp_path):
    import bench

    bg = bench.BackgroundGen("phased10k", 40, 1, str(tmp_path))
    try:
        assert bg.p.poll() is None
        bg.pause()
        time.sleep(0.3)
        assert _state(bg.p.pid) in ("T", "t"), _state(bg.p.pid)  # stopped: not runnable, not sleeping
        cpu0 = sum(int(x) for x in open(f"/proc/{bg.p.pid}/stat").read().rsplit(")", 1)[1].split()[11:13])
        time.sleep(0.5)
        cpu1 = sum(int(x) for x in open(f"/proc/{bg.p.pid}/stat").read().rsplit(")", 1)[1].split()[11:13])
        assert cpu1 == cpu0  # not a tick of CPU time while stopped
        bg.pause()  # idempotent
        bg.resume()
        bg.resume()
        assert bg.wait(timeout=300)
        assert bg.paused_s >= 0.7 and bg.seconds > bg.paused_s
        assert os.path.getsize(bg.prefix + ".bam") > 100_000 and os.path.exists(bg.prefix + ".bam.bai") and os.path.exists(bg.prefix + ".bed")
        assert len(open(bg.prefix + ".bed").read().splitlines()) == 40
    finally:
        if bg.p.poll() is None:
            bg.abort()


def test_rest_helper_leaves_the_writer_stopped(tmp_path, monkeypatch):
    import bench

    bg = bench.BackgroundGen("phased10k", 40, 1, str(tmp_path))
    monkeypatch.setattr(bench, "BG", bg)
    try:
        bench.rest_then_quiet(0.4)
        if bg.p.poll() is None:  # (a writer that was already through has nothing to stop)
            assert _state(bg.p.pid) in ("T", "t")
        bench.bg_resume()
        assert bg.wait(timeout=300)
    finally:
        if bg.p.poll() is None:
            bg.abort()


def test_host_cores_available_is_the_quota_not_the_machine():
    import bench

    n = bench.host_cores_available()
    assert 1 <= n <= len(os.sched_getaffinity(0))
    from inquistr_amd import call

    assert n == call.load().inq_host_granted_cpus()  # bench.py's CPU legs and the library's reader pools count the same cores
