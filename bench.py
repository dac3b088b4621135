#!/usr/bin/env python3
"""bench.py — loci/s of the `inquiSTR call` hot path on MI355X.

One step = one pass of the hot path (every locus of one synthetic batch -> two medians) with
the batch already resident in HBM.  N GPUs = N processes (torch.distributed, backend nccl = RCCL),
loci sharded by rank with no data-path collective; the only exchange is the gather of the
per-shard result rows to rank 0, overlapped with the next step on a second stream.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def host_cores_available() -> int:
    """Host cores this process may really use, uncapped: the affinity mask, cut by the cgroup CPU quota if one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                q, p = txt
            else:
                q, p = txt[0], open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().split()[0]
            if q not in ("max", "-1"):
                n = min(n, max(1, int(q) // int(p)))
            break
        except Exception:
            continue
    return max(1, n)


def host_threads() -> int:
    """The product's own -t and the CPU legs' default thread count: the available cores capped at 16 (the per-GPU CPU
    share of the measurement box; INQ_CPU_THREADS overrides).  CPU mode B is ALSO timed on host_cores_available() threads
    when that is more (the reference's -t >= 2 runs on rayon's global pool = all cores, src/call.rs:104-118)."""
    return max(1, min(host_cores_available(), int(os.environ.get("INQ_CPU_THREADS", "16"))))


def cpu_baseline(wl, sample_loci: int, budget_s: float = 10.0):
    """The oracle (CPU restatement, kind "port") on the first `sample_loci` loci of the same
    workload, all host cores (OpenMP over loci = the reference's rayon par_bridge)."""
    from inquistr_amd import synth
    from oracle import orc

    orc.build()
    threads = host_threads()
    batch = synth.generate_numpy(wl, 0, sample_loci)
    # repeat the sample until about 10 s of wall time (x `threads` cores of CPU work) have gone by
    times = []
    t_all = time.perf_counter()
    res = None
    while len(times) < 5 or (time.perf_counter() - t_all < budget_s and len(times) < 2000):
        t0 = time.perf_counter()
        code, res = orc.call_batch(batch, threads=threads)
        times.append(time.perf_counter() - t0)
        assert code == 0
    best, med = min(times), sorted(times)[len(times) // 2]
    return res, {
        "value": sample_loci / med,
        "unit": "loci/s",
        "cores": threads,
        "kind": "port",
        "best": sample_loci / best,
        "sample": f"first {sample_loci} loci of {wl.name} ({batch.n_pairs} reads, {int(batch.cigar_ops_per_pair().sum())} CIGAR ops), "
        f"SoA already in host memory, {len(times)} repetitions over {sum(times):.1f} s of wall time on {threads} threads, "
        f"median {med * 1e3:.1f} ms (best {best * 1e3:.1f} ms); ARITHMETIC ONLY (no BGZF inflate, no BAM record decode: "
        "not a speed-up over `inquiSTR call`, see the l2 block for that), CPU restatement of the reference, not the Rust binary",
    }


PCIE_SPEC_GBS = 64.0  # PCIe Gen5 x16, one direction, before protocol overhead (the link an MI355X hangs on)
# Seconds of rest in front of every TIMED process that uses the GPU.  When a process that held GBs of device memory leaves, the
# driver wipes that memory and takes its queues apart for some tenths of a second, and a process that starts meanwhile waits
# for it inside hipInit (0.10 s -> 0.2 - 0.4 s) and inside its allocations (50 ms per GB): the whole "start-up varies four-fold"
# of rounds 2 - 3 was the PREVIOUS run of the loop (profiles/r04_results/back_to_back_vs_paused_processes.txt: twelve runs with
# 1.5 s between them 0.156 - 0.162 s each but three, the same runs back to back 0.16 - 0.77 s).  The timed runs below therefore
# each meet an idle device - what somebody who runs one command sees -, and `seconds_back_to_back` keeps three runs without the rest.
GPU_REST_S = float(os.environ.get("INQ_BENCH_REST_S", "1.2"))


class BackgroundGen:
    """The large SEQ-bearing BAM of the l2_seq_large block, written by a child process (tools/make_synth_bam.py, niced) from the first
    second of the run - 31 GB at zlib level 6 are ~2 core-hours/60 - and STOPPED (SIGSTOP to its process group) during every timed
    region of this script: it makes progress while this process generates inputs, runs the profiler's child passes, and above all
    during the rests in front of every timed GPU process (1.2 s each, some fifty of them).  r4 wrote the file in 114 s of a 303 s run
    with everything else waiting."""

    def __init__(self, workload: str, loci: int, level: int, directory: str):
        import signal

        self._sig = signal
        self.loci, self.level = loci, level
        self.prefix = os.path.join(directory, f"{workload}_{loci}")
        self.t0 = time.perf_counter()
        self.paused_s, self._t_pause = 0.0, None
        self.log = open(self.prefix + ".gen.log", "w")
        # two threads fewer than the cores the cgroup grants: a process group that uses its whole CPU quota is throttled for the rest of
        # the scheduler's 100 ms period, and the timed process that starts right behind a rest would begin inside that hole
        cmd = [sys.executable, os.path.join(ROOT, "tools", "make_synth_bam.py"), workload, str(loci), self.prefix, "native-seq", str(level),
               str(max(1, host_cores_available() - 2))]
        self.p = subprocess.Popen(cmd, stdout=self.log, stderr=subprocess.STDOUT, start_new_session=True, preexec_fn=lambda: os.nice(19), cwd=ROOT)
        self.seconds = None

    def _kill(self, sig):
        try:
            os.killpg(self.p.pid, sig)
        except (ProcessLookupError, PermissionError):
            pass

    def pause(self):
        if self.p.poll() is None and self._t_pause is None:
            self._kill(self._sig.SIGSTOP)
            self._t_pause = time.perf_counter()

    def resume(self):
        if self._t_pause is not None:
            self.paused_s += time.perf_counter() - self._t_pause
            self._t_pause = None
            self._kill(self._sig.SIGCONT)

    def wait(self, timeout: float):
        """True when the file is there; the child gets the machine to itself for what is left."""
        self.resume()
        try:
            rc = self.p.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            self.abort()
            return False
        if self.seconds is None:
            self.seconds = time.perf_counter() - self.t0
        self.log.close()
        return rc == 0 and os.path.exists(self.prefix + ".bam")

    def abort(self):
        self._kill(self._sig.SIGCONT)
        self._kill(self._sig.SIGKILL)
        try:
            self.p.wait(timeout=30)
        except Exception:  # noqa: BLE001
            pass


BG = None  # the running BackgroundGen, if any


def bg_pause():
    if BG is not None:
        BG.pause()


def bg_resume():
    if BG is not None:
        BG.resume()


def rest_then_quiet(seconds: float):
    """The rest in front of a timed GPU process: the background generator works through it and is stopped when it ends."""
    bg_resume()
    if seconds > 0.2:
        time.sleep(seconds - 0.15)
    bg_pause()
    time.sleep(min(0.15, max(seconds, 0.0)))  # a scheduler period without the generator in front of the timed process


def h2d_copy_peak(dev, mb: int = 256, reps: int = 6):
    """What the host-to-device link delivers on this box, measured in this run: one pinned buffer of a span's size (256 MB) copied
    to the device `reps` times on a stream of its own, HIP events around each copy, best and median.  The `pcie` roofline objects
    of the L2 blocks are priced against it (and against the 64 GB/s of the Gen5 x16 specification)."""
    import statistics

    import torch

    n = mb << 20
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    s = torch.cuda.Stream(device=dev)
    rates = []
    with torch.cuda.stream(s):
        d.copy_(h, non_blocking=True)  # the first copy sets the path up
        s.synchronize()
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            d.copy_(h, non_blocking=True)
            e1.record(s)
            s.synchronize()
            rates.append(n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del h, d
    return {"value": max(rates), "median": statistics.median(rates), "unit": "GB/s",
            "what": f"pinned host -> device copy of {mb} MB, best of {reps} (HIP events), measured in this run"}


def startup_floor(reps: int = 3):
    """What ANY process that launches one kernel on this box pays: tools/hip_startup_probe.hip (hipInit, a stream, one empty kernel,
    three allocations, one small copy - nothing of this repo), whole process from start to exit, measured in this run.  Printed
    beside every whole-process time of the L2 blocks: the difference is what the product adds."""
    import statistics

    exe = os.path.join(ROOT, "inquistr_amd", "lib", "hip_startup_probe")
    if not os.path.exists(exe):
        return None
    ts, stages = [], None
    for _ in range(reps):
        rest_then_quiet(GPU_REST_S)
        t = time.perf_counter()
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        ts.append(time.perf_counter() - t)
        if r.returncode != 0:
            return None
        stages = r.stdout
    inner = 0.0
    for ln in (stages or "").splitlines():
        f = ln.split()
        if len(f) >= 2 and f[-1] == "ms":
            try:
                inner += float(f[-2])
            except ValueError:
                pass
    med = statistics.median(ts)
    return {"seconds_median": med, "seconds_all": ts, "runs": reps, "rest_before_each_run_s": GPU_REST_S, "inside_main_ms_last_run": inner,
            "start_and_exit_ms_last_run": max(0.0, ts[-1] * 1e3 - inner),
            "what": "bare HIP process (tools/hip_startup_probe.hip: hipInit, one stream, one empty kernel, 6 GB of allocations, a 1 MB copy), "
                    "start to exit; `inside_main` = the sum of its own stage clocks, the rest is loading the runtime's libraries and the exit"}


def l1_block(wl, dev_index: int, loci: int = 50_000, reps: int = 5):
    """L1: the host-buffer entry inq_call_batch - H2D of the SoA, the locus kernels, D2H of the rows - on pinned host
    buffers; bound by the PCIe link, never `value`."""
    import ctypes as C

    import numpy as np

    from inquistr_amd import hipcall, synth
    from inquistr_amd.batch import Batch

    b = synth.generate_numpy(wl, 0, loci)
    L = hipcall.load()
    keep, arrs = [], {}
    try:
        for name in ("cigar", "reads", "pair_read", "locus_pair_off", "locus_start", "locus_end"):
            arr = getattr(b, name)
            ptr = C.c_void_p()
            if L.inq_alloc_pinned(max(arr.nbytes, 1), C.byref(ptr)) != 0:
                raise RuntimeError("inq_alloc_pinned failed")
            keep.append(ptr)
            buf = (C.c_uint8 * max(arr.nbytes, 1)).from_address(ptr.value)
            out = np.frombuffer(buf, dtype=np.uint8, count=arr.nbytes).view(arr.dtype)
            out[...] = arr
            arrs[name] = out
        pb = Batch(minlen=b.minlen, support=b.support, unphased=b.unphased, **arrs)
        with hipcall.Context(dev_index) as ctx:
            ctx.call_batch(pb)  # allocations
            times = []
            for _ in range(reps):
                t0 = time.perf_counter()
                rc, _res = ctx.call_batch(pb)
                times.append(time.perf_counter() - t0)
                if rc != 0:
                    raise RuntimeError(f"inq_call_batch returned {rc}")
        best = min(times)
        nbytes = b.cigar.nbytes + b.reads.nbytes + b.pair_read.nbytes + b.locus_pair_off.nbytes + b.locus_start.nbytes + b.locus_end.nbytes
        return {"level": "L1: inq_call_batch on pinned host buffers (H2D + kernels + D2H)", "loci": loci, "loci_per_s": loci / best,
                "ms_per_call": best * 1e3, "ms_all": [x * 1e3 for x in times], "host_bytes_in": int(nbytes), "GBps_host_to_device_incl_kernels": nbytes / best / 1e9,
                "note": "PCIe-inclusive: never `value`"}
    finally:
        for ptr in keep:
            L.inq_free_pinned(ptr)


def l2_block(workload: str, loci: int, threads: int, device, reps: int = 5, a_loci: int = 10_000, c_loci: int = 2_000, seq: bool = False,
             level: int = 6, lean: bool = False, served_callers: int = 0, cohort: bool = True, floor=None, h2d=None, trace_runs: int = 1,
             ready_prefix: str = "", ready_gen_s: float = 0.0, b2b_runs: int = 3):
    """End to end (BAM + BED -> .inq) next to the reference-shaped CPU programs, small enough for the default run.
    Product CLI with the device front end: median of `reps` whole-process wall times (HIP start-up included; the CLI
    leaves through _Exit once the rows are written).  CPU side = oracle/ref_shaped_call, the reference's control flow
    (B = one reader per worker, A = BAM + .bai re-opened per locus as src/call.rs:217 does with -t >= 2,
    C = serial) around the CPU restatement - not the Rust binary.  A and C run on the first a_loci / c_loci targets of the
    same BAM (A costs ~0.5 ms per locus and core), B on all of them - at `threads` threads and, when the box offers more, at
    host_cores_available() threads too; outputs are compared byte for byte.
    seq: records shaped like a real long-read BAM (SEQ + QUAL of the query length, NM, ML / MM tags, HP last: ~18 KB per
    record instead of ~0.85 KB; tools/synth_bam_writer.cc inq_synth_write_bam_seq), the shape north_star's ">= 10x the
    reference CPU inquiSTR call on a synthetic long-read BAM" is about.
    lean: a file of many GB - only the product CLI (device front end), CPU mode B on all granted cores and the stage times.
    level: zlib level of the BGZF blocks (6 = htslib's default for BAM output; 1 = round 1 - 3's files).
    served_callers: > 0 adds that many `inquistr call` processes queueing at one resident server (off in the default line: with
    this process and the server they were six processes on the card).  cohort: the many-files-in-one-process figures.
    floor / h2d: this run's startup_floor() and h2d_copy_peak(), quoted beside the whole-process times / in the `pcie` object."""
    import statistics
    import subprocess
    import tempfile

    from inquistr_amd import synth
    from tools import make_synth_bam

    wl = synth.WORKLOADS[workload]
    tmp = os.path.dirname(ready_prefix) if ready_prefix else tempfile.mkdtemp(prefix="inq_l2_")
    prefix = ready_prefix or os.path.join(tmp, f"{workload}_{loci}")
    try:
        t0 = time.perf_counter()
        info = {}
        if ready_prefix:  # written beside the earlier blocks (BackgroundGen)
            gen_s = ready_gen_s
        else:
            bg_pause()  # (the cgroup grants a fixed number of cores: two writers at once only halve each other)
            make_synth_bam.write_native(workload, loci, prefix, threads=host_cores_available(), device=device, seq=seq, level=level, info=info)
            gen_s = time.perf_counter() - t0
        bg_pause()
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref_shaped_call"], stdout=subprocess.DEVNULL)
        cli = os.path.join(ROOT, "inquistr_amd", "lib", "inquistr")
        ref = os.path.join(ROOT, "oracle", "ref_shaped_call")
        un = ["-u"] if wl.unphased else []
        bed_lines = open(prefix + ".bed").read().splitlines(keepends=True)

        def sub_bed(n):
            p = f"{prefix}.first{n}.bed"
            open(p, "w").write("".join(bed_lines[:n]))
            return p

        def run(cmd, env=None, timeout=None, rest=0.0):
            if rest:
                rest_then_quiet(rest)
            t = time.perf_counter()
            r = subprocess.run(cmd, capture_output=True, env=env, timeout=timeout)
            dt = time.perf_counter() - t
            if r.returncode != 0:
                raise RuntimeError(f"{cmd[0]} exited {r.returncode}: {r.stderr.decode()[-300:]}")
            return dt, r.stdout

        cmd = [cli, "call", prefix + ".bam", "-R", prefix + ".bed", "-t", str(threads), "--sample-name", "S"] + un
        # Page cache + code objects warm, as for every program below - and warm means read TWICE: the second read of a freshly written
        # file moves every one of its page-cache pages to the active LRU list (mark_page_accessed: /proc/vmstat pgactivate + 3 131 362
        # for the 12.8 GB file, 0 - 70 000 in every other run), under one lock per memory node, and sixteen readers then burn 13 - 15 s
        # of CPU time instead of 3 on that lock: reads of 13 - 21 ms per span instead of 4, always run 2, never run 1 or 3 + - that was
        # round 3's "one slow run in five" (profiles/r04_results/slow_second_read.txt).  A kernel matter, not a property of any program here.
        run(cmd, dict(os.environ, INQ_FRONTEND="device"))
        run(cmd, dict(os.environ, INQ_FRONTEND="device"))
        dev = [run(cmd, dict(os.environ, INQ_FRONTEND="device"), rest=GPU_REST_S) for _ in range(reps)]
        t_dev = statistics.median(t for t, _ in dev)
        out_dev = dev[0][1]
        b2b = [run(cmd, dict(os.environ, INQ_FRONTEND="device")) for _ in range(b2b_runs)]  # ... and without the rest, each behind the last one's exit
        t_host, out_host = (None, None) if lean else run(cmd, dict(os.environ, INQ_FRONTEND="host"), rest=GPU_REST_S)
        rows = out_dev.splitlines(keepends=True)
        bam_bytes = os.path.getsize(prefix + ".bam")
        res = {
            "level": "L2: BAM + BED -> .inq, whole process, start to exit", "workload": workload, "loci": loci, "threads": threads,
            "host_cores_available": host_cores_available(),
            "bam_mb": bam_bytes / 1e6, "bam_gen_s": gen_s, "zlib_level": level,
            "records": ("SEQ + QUAL of the query length, NM:i, ML:B,C + MM:Z, HP:C last (long-read record shape, ~18 KB per record; "
                        "bases ACGT, Phred a clamped random walk)" if seq else "SEQ '*' (CIGAR-only records), HP:C"),
            **({"inflated_mb": info["inflated_bytes"] / 1e6, "bgzf_blocks": info["n_blocks"]} if info else {}),
            "gpu_cli_device_front": {"seconds_median": t_dev, "seconds_all": [t for t, _ in dev], "runs": reps, "rest_before_each_run_s": GPU_REST_S,
                                     "seconds_back_to_back": [t for t, _ in b2b],
                                     "loci_per_s": loci / t_dev, "identical_across_runs": all(o == out_dev for _, o in dev + b2b)},
        }
        if not lean:
            res["gpu_cli_host_front"] = {"seconds": t_host, "loci_per_s": loci / t_host, "inq_identical": out_host == out_dev}
        args_tail = [str(int(wl.unphased)), str(wl.minlen), str(wl.support), "S"]
        tb, out_b = run([ref, prefix + ".bam", prefix + ".bed", "B", str(threads)] + args_tail)
        res["cpu_B"] = {"seconds": tb, "loci": loci, "loci_per_s": loci / tb, "cores": threads, "inq_identical": out_b == out_dev}
        all_cores = host_cores_available()
        if all_cores > threads:  # the reference's -t >= 2 uses every core (src/call.rs:104-118): mode B there as well
            tb2, out_b2 = run([ref, prefix + ".bam", prefix + ".bed", "B", str(all_cores)] + args_tail)
            res["cpu_B_all_cores"] = {"seconds": tb2, "loci": loci, "loci_per_s": loci / tb2, "cores": all_cores, "inq_identical": out_b2 == out_dev}
        # the device front end's own stage times for this file (HIP events / host clocks inside the CLI, one extra run)
        rest_then_quiet(GPU_REST_S)
        r = subprocess.run(cmd, capture_output=True, env=dict(os.environ, INQ_FRONTEND="device", INQ_TIMING="2"))
        res["device_front_stages"] = stage_summary(r.stderr.decode(), bam_bytes)
        loops = [res["device_front_stages"].get("span_loop_s")]
        for _ in range(max(0, trace_runs - 1)):  # more samples of the span loop's own time (INQ_TIMING=1: one line per run)
            rest_then_quiet(GPU_REST_S)
            r2 = subprocess.run(cmd, capture_output=True, env=dict(os.environ, INQ_FRONTEND="device", INQ_TIMING="1"))
            loops.append(stage_summary(r2.stderr.decode(), bam_bytes).get("span_loop_s"))
        loops = [x for x in loops if x]
        if loops:
            # The L2 path is bound by the host-to-device link: every compressed byte of the spans crosses it once, and nothing
            # else of comparable size does.  achieved = compressed bytes of the spans / the span loop's own time (first span
            # handed to the device -> last flush: the process's fixed costs are in `startup_floor`, not here).
            comp_gb = res["device_front_stages"]["span_loop_comp_mb"] / 1e3
            ach = comp_gb / statistics.median(loops)
            res["pcie"] = {"bound": "pcie", "achieved": ach, "unit": "GB/s", "peak": PCIE_SPEC_GBS, "frac": ach / PCIE_SPEC_GBS,
                           "span_loop_s_all": loops, "achieved_all": [comp_gb / x for x in loops], "compressed_gb": comp_gb,
                           "what": "compressed bytes of the spans / time from the first span's device call to the last flush (CLI's own clock, INQ_TIMING), median"}
            if h2d:
                res["pcie"]["peak_measured"] = h2d
                res["pcie"]["frac_of_measured"] = ach / h2d["value"]
        if floor:
            res["startup_floor"] = floor
            res["gpu_cli_device_front"]["seconds_median_minus_floor"] = t_dev - floor["seconds_median"]
        # the same command, handed to a process that already holds the device context (`inquistr serve`; `inquistr call` with
        # INQ_SERVER set passes its arguments and its stdout to it): what a file costs in a pipeline that starts one process per
        # sample, without the HIP runtime's start-up and the exit of a process that mapped GBs.  Whole client process, start to exit.
        try:
            sock = os.path.join(tmp, "inq.sock")
            server = subprocess.Popen([cli, "serve", "--socket", sock, "--idle-exit", "300"], env=dict(os.environ, INQ_FRONTEND="device"),
                                      stderr=subprocess.DEVNULL)
            try:
                for _ in range(400):
                    if os.path.exists(sock):
                        break
                    time.sleep(0.025)
                if not os.path.exists(sock) or server.poll() is not None:
                    # (never fall through to callers that would each start a context of their own: the box allows six on the card)
                    raise RuntimeError("inquistr serve did not come up")
                env_s = dict(os.environ, INQ_SERVER=sock)
                run(cmd, env_s, timeout=300)  # the context's start-up is the first caller's (a stuck server must not hold the line)
                sv = [run(cmd, env_s, timeout=300) for _ in range(reps)]
                t_sv = statistics.median(t for t, _ in sv)
                res["gpu_cli_served"] = {"seconds_median": t_sv, "seconds_all": [t for t, _ in sv], "runs": reps, "loci_per_s": loci / t_sv,
                                         "inq_identical": all(o == out_dev for _, o in sv),
                                         "note": "inquistr call with INQ_SERVER=<socket of a running `inquistr serve`>: same CLI, the device context is resident"}
                res["speedup_served_vs_B"] = res["gpu_cli_served"]["loci_per_s"] / res["cpu_B"]["loci_per_s"]
                if served_callers > 0:
                    # ... and with callers queueing (a workflow manager starts several at once): the server stages file k + 1 while it
                    # calls file k; 4 callers started together, wall time until the last has left
                    n_par = served_callers
                    if server.poll() is not None:
                        raise RuntimeError("inquistr serve is gone")
                    t = time.perf_counter()
                    # (stdout into files: the server writes a caller's rows while the others wait their turn; pipes read one after the
                    # other by this process would fill up and stall the queue)
                    fs = [open(os.path.join(tmp, f"par{i}.inq"), "wb") for i in range(n_par)]
                    ps = [subprocess.Popen(cmd, env=env_s, stdout=fs[i], stderr=subprocess.DEVNULL) for i in range(n_par)]
                    try:
                        for p in ps:
                            p.wait(timeout=300)
                    except Exception:
                        for p in ps:
                            if p.poll() is None:
                                p.kill()
                        raise
                    dt_par = time.perf_counter() - t
                    for f in fs:
                        f.close()
                    outs = [open(os.path.join(tmp, f"par{i}.inq"), "rb").read() for i in range(n_par)]
                    res["gpu_cli_served"]["callers_at_once"] = {"callers": n_par, "seconds_all_done": dt_par, "seconds_per_file": dt_par / n_par,
                                                                 "loci_per_s": loci * n_par / dt_par, "inq_identical": all(o == out_dev for o in outs),
                                                                 "speedup_vs_B": (loci * n_par / dt_par) / res["cpu_B"]["loci_per_s"]}
            finally:
                if server.poll() is None:
                    subprocess.run([cli, "serve", "--socket", sock, "--quit"], capture_output=True, timeout=60)
                    try:
                        server.wait(timeout=60)
                    except Exception:  # noqa: BLE001
                        pass
                if server.poll() is None:
                    server.kill()
                    server.wait(timeout=60)
        except Exception as e:  # noqa: BLE001
            res["gpu_cli_served"] = {"error": f"{type(e).__name__}: {e}"}
        if lean:
            res["inq_identical"] = bool(res["cpu_B"]["inq_identical"] and res.get("cpu_B_all_cores", res["cpu_B"])["inq_identical"])
            res["speedup_vs_B"] = res["gpu_cli_device_front"]["loci_per_s"] / res["cpu_B"]["loci_per_s"]
            if "cpu_B_all_cores" in res:
                res["speedup_vs_B_all_cores"] = res["gpu_cli_device_front"]["loci_per_s"] / res["cpu_B_all_cores"]["loci_per_s"]
            res["note"] = ("speed-up = ratio of loci/s; cpu_B = oracle/ref_shaped_call mode B (one reader per worker, own BGZF / BAM / BAI reader, "
                           "no code shared with the product), not the Rust binary")
            return res
        na, nc = min(a_loci, loci), min(c_loci, loci)
        ta, out_a = run([ref, prefix + ".bam", sub_bed(na), "A", str(threads)] + args_tail)
        res["cpu_A"] = {"seconds": ta, "loci": na, "loci_per_s": na / ta, "cores": threads,
                        "inq_identical": out_a == b"".join(rows[: na + 1])}
        tc, out_c = run([ref, prefix + ".bam", sub_bed(nc), "C", "1"] + args_tail)
        res["cpu_C"] = {"seconds": tc, "loci": nc, "loci_per_s": nc / tc, "cores": 1,
                        "inq_identical": sorted(out_c.splitlines()) == sorted(b"".join(rows[: nc + 1]).splitlines())}
        res["inq_identical"] = bool(res["cpu_B"]["inq_identical"] and res["cpu_A"]["inq_identical"] and res["cpu_C"]["inq_identical"]
                                    and res["gpu_cli_host_front"]["inq_identical"])
        for m in "BAC":
            res[f"speedup_vs_{m}"] = res["gpu_cli_device_front"]["loci_per_s"] / res[f"cpu_{m}"]["loci_per_s"]
        # a cohort: the same command for several BAMs in ONE process (`inquistr cohort` = inq_session_call_many: one HIP context,
        # file k + 1 staged while file k is called).  The HIP runtime's start-up - 0.1 to 0.4 s from run to run, most of a single
        # file's time at this size - is paid once; what an added file costs is (t(n files) - t(1 file)) / (n - 1).
        if cohort:
            try:
                n_co = 5
                links = []
                for k in range(n_co):
                    ln = f"{prefix}.co{k}.bam"
                    for ext in ("", ".bai"):
                        if os.path.exists(ln + ext):
                            os.unlink(ln + ext)
                        os.link(prefix + ".bam" + ext, ln + ext)
                    links.append(ln)
                outdir = os.path.join(tmp, "cohort_out")
                os.makedirs(outdir, exist_ok=True)
                co = [cli, "cohort", "-R", prefix + ".bed", "-t", str(threads), "--out-dir", outdir] + un
                env_d = dict(os.environ, INQ_FRONTEND="device")
                t1 = statistics.median(run(co + links[:1], env_d, rest=GPU_REST_S)[0] for _ in range(3))
                tn = statistics.median(run(co + links, env_d, rest=GPU_REST_S)[0] for _ in range(3))
                per = max((tn - t1) / (n_co - 1), 1e-9)
                # the library's own clock between two files of the cohort (stderr of one more run): what a file costs once the
                # context is there, without the process's start and its exit (tearing down GBs of mappings: tenths of a second)
                r = subprocess.run(co + links, capture_output=True, env=dict(env_d, INQ_TIMING="1"))
                import re as _re

                inner = [float(x) for x in _re.findall(r"\[inq session\].*?([\d.]+) ms since the previous file finished", r.stderr.decode())]
                body = out_dev.split(b"\n", 1)[1]
                same = all(open(os.path.join(outdir, os.path.basename(ln)[: -len(".bam")] + ".inq"), "rb").read().split(b"\n", 1)[1] == body for ln in links)
                res["cohort"] = {"files": n_co, "seconds_1_file": t1, "seconds_n_files": tn, "seconds_per_added_file": per,
                                 "loci_per_s_per_added_file": loci / per, "speedup_vs_B_per_added_file": (loci / per) / res["cpu_B"]["loci_per_s"],
                                 "ms_per_file_inside_the_session": inner[1:], "loci_per_s_inside_the_session": (loci / (statistics.median(inner[1:]) / 1e3)) if len(inner) > 1 else None,
                                 "rows_identical_to_single_calls": bool(same),
                                 "note": "inquistr cohort (many calls in one process, one device context); medians of 3 whole-process wall times; the files are hard links of the one BAM (every program here reads from the page cache)"}
            except Exception as e:  # noqa: BLE001
                res["cohort"] = {"error": f"{type(e).__name__}: {e}"}
        if "cpu_B_all_cores" in res:
            res["inq_identical"] = bool(res["inq_identical"] and res["cpu_B_all_cores"]["inq_identical"])
            res["speedup_vs_B_all_cores"] = res["gpu_cli_device_front"]["loci_per_s"] / res["cpu_B_all_cores"]["loci_per_s"]
        res["note"] = ("speed-ups are ratios of loci/s; cpu_* = oracle/ref_shaped_call (CPU restatement in the reference's control "
                       "flow with its own BGZF / BAM / BAI reader, oracle/minibam.h: no code shared with the product), not the Rust binary; A and C timed on a prefix of the targets")
        return res
    finally:
        import shutil

        shutil.rmtree(tmp, ignore_errors=True)
        bg_resume()  # behind the block's timed regions the large file's writer goes on


def stage_summary(stderr_text: str, bam_bytes: int):
    """Sums the per-span stage lines the CLI prints with INQ_TIMING=2 ("[inq span] ... | upload U inflate I scan S join J call C
    ms | wall W ms": HIP-event times of the device stages) into per-file figures, the inflate's rate among them."""
    import re

    tot = {"spans": 0, "comp_mb": 0.0, "inflated_mb": 0.0, "upload_ms": 0.0, "inflate_ms": 0.0, "scan_ms": 0.0, "join_ms": 0.0, "call_ms": 0.0,
           "wall_ms": 0.0, "loci_per_span": [], "loci_per_call_launch": []}
    pat = re.compile(r"\[inq span\].*?loci (\d+) comp ([\d.]+) MB -> ([\d.]+) MB.*upload ([\d.]+) inflate ([\d.]+) scan ([\d.]+) join ([\d.]+) ms \| wall ([\d.]+) ms")
    pat_call = re.compile(r"\[inq call\].*? (\d+) loci, ([\d.]+) MB of CIGARs: locus kernels ([\d.]+) ms \| wall ([\d.]+) ms")
    for ln in stderr_text.splitlines():
        m = pat.search(ln)
        if m:
            v = [float(x) for x in m.groups()]
            tot["spans"] += 1
            tot["loci_per_span"].append(int(v[0]))
            for k, x in zip(("comp_mb", "inflated_mb", "upload_ms", "inflate_ms", "scan_ms", "join_ms", "wall_ms"), v[1:]):
                tot[k] += x
        m = pat_call.search(ln)
        if m:  # the locus kernels run over the batches of several spans at once (inq_call_flush)
            tot["loci_per_call_launch"].append(int(m.group(1)))
            tot["call_ms"] += float(m.group(3))
            tot["wall_ms"] += float(m.group(4))
    if tot["spans"] and tot["inflate_ms"] > 0:
        tot["inflate_in_GBps"] = tot["comp_mb"] / tot["inflate_ms"]   # MB / ms = GB / s
        tot["inflate_out_GBps"] = tot["inflated_mb"] / tot["inflate_ms"]
        tot["note"] = "inflate_ms includes the CRC check of every block; rates = compressed bytes in / inflated bytes out per second of it"
    for ln in stderr_text.splitlines():
        if ln.startswith("[inq timing] device front end:"):
            tot["cli_timing"] = ln[len("[inq timing] "):]
        m = re.search(r"\[inq timing\] span loop: (\d+) spans, ([\d.]+) MB compressed, ([\d.]+) s", ln)
        if m:  # first span handed to the device -> last flush done: the file's cost behind the process's fixed costs
            tot["span_loop_s"] = float(m.group(3))
            tot["span_loop_comp_mb"] = float(m.group(2))
        m = re.search(r"device context ready \(inq_ctx_create ([\d.]+) ms\)", ln)
        if m:
            tot["ctx_create_ms"] = float(m.group(1))
    return tot


def l2_end_to_end(workload: str, loci: int, threads: int, reps: int, keep: str = "", cpu_modes: str = "CBA", seq: bool = False):
    """L2: BAM + BED -> .inq.  The product CLI (C++ sweep front end + HIP kernels) next to this bench's CPU
    baseline leg at that level: oracle/ref_shaped_call, the reference's control flow (BASELINE.md §3 modes
    A / B / C) around the oracle — a CPU restatement, not the Rust binary.  Outputs are compared byte for
    byte.  Returns one JSON-able dict."""
    import subprocess
    import tempfile

    from inquistr_amd import synth
    from tools import make_synth_bam

    wl = synth.WORKLOADS[workload]
    tmp = keep or tempfile.mkdtemp(prefix="inq_l2_")
    prefix = os.path.join(tmp, f"{workload}_{loci}" + ("_seq" if seq else ""))
    t0 = time.time()
    if not os.path.exists(prefix + ".bam"):
        if seq:
            make_synth_bam.write(workload, loci, prefix, seq=True)  # SEQ / QUAL records: Python writer only
        else:
            import torch

            make_synth_bam.write_native(workload, loci, prefix, threads=threads,
                                        device=torch.device("cuda:0") if torch.cuda.is_available() else None)
    gen_s = time.time() - t0
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref_shaped_call"], stdout=subprocess.DEVNULL)
    cli = os.path.join(ROOT, "inquistr_amd", "lib", "inquistr")
    ref = os.path.join(ROOT, "oracle", "ref_shaped_call")
    un = ["-u"] if wl.unphased else []

    def timed(cmd, n, env=None):
        best, out = None, None
        for _ in range(n):
            t = time.perf_counter()
            r = subprocess.run(cmd, capture_output=True, env=env)
            dt = time.perf_counter() - t
            if r.returncode != 0:
                raise SystemExit(f"{cmd} failed: {r.stderr.decode()[-500:]}")
            best = dt if best is None else min(best, dt)
            out = r.stdout
        return best, out

    res = {}
    cmd = [cli, "call", prefix + ".bam", "-R", prefix + ".bed", "-t", str(threads), "--sample-name", "S"] + un
    t_host, out_host = timed(cmd, reps + 1, dict(os.environ, INQ_FRONTEND="host"))
    t_dev, out_dev = timed(cmd, reps + 1, dict(os.environ, INQ_FRONTEND="device"))
    res["gpu_cli_host_front"] = {"seconds": t_host, "loci_per_s": loci / t_host, "threads": threads}
    res["gpu_cli_device_front"] = {"seconds": t_dev, "loci_per_s": loci / t_dev, "threads": threads,
                                   "inq_identical_to_host_front": out_dev == out_host}
    r = subprocess.run(cmd, capture_output=True, env=dict(os.environ, INQ_FRONTEND="device", INQ_TIMING="2"))
    res["device_front_stages"] = [ln for ln in r.stderr.decode().splitlines() if ln.startswith("[inq")]
    for k, v in (("INQ_SPAN_MB", "64"), ("INQ_SPAN_MB", "256")):  # variants of the host half
        tv, outv = timed(cmd, reps, dict(os.environ, INQ_FRONTEND="device", **{k: v}))
        res[f"gpu_cli_device_front_{k}={v}"] = {"seconds": tv, "inq_identical_to_host_front": outv == out_host}
    t_auto, out_auto = timed(cmd, reps, dict(os.environ))
    res["gpu_cli_auto_front"] = {"seconds": t_auto, "inq_identical_to_host_front": out_auto == out_host}
    t_gpu, out_gpu = min((t_host, out_host), (t_dev, out_dev), key=lambda x: x[0])
    res["gpu_cli"] = {"seconds": t_gpu, "loci_per_s": loci / t_gpu, "threads": threads,
                      "front_end": "device" if t_dev <= t_host else "host"}
    for mode, thr in (("C", 1), ("B", threads), ("A", threads)):
        if mode not in cpu_modes:
            continue
        t, out = timed([ref, prefix + ".bam", prefix + ".bed", mode, str(thr), str(int(wl.unphased)), str(wl.minlen),
                        str(wl.support), "S"], 1 if mode == "A" else reps)
        same = sorted(out.splitlines()) == sorted(out_gpu.splitlines()) if mode == "C" else out == out_gpu
        res[f"cpu_{mode}"] = {"seconds": t, "loci_per_s": loci / t, "threads": thr, "inq_identical": bool(same)}
    return {"level": "L2 end-to-end BAM+BED -> .inq", "workload": workload, "loci": loci,
            "records": "SEQ + QUAL + ML/MM tags, HP last (real long-read record shape)" if seq else "SEQ '*' (CIGAR-only records)",
            "bam_mb": os.path.getsize(prefix + ".bam") / 1e6, "bam_gen_s": gen_s, **res,
            **{f"speedup_vs_{m}": res[f"cpu_{m}"]["seconds"] / t_gpu for m in "ABC" if f"cpu_{m}" in res},
            "note": "CPU modes = oracle/ref_shaped_call: CPU restatement of the reference's control flow, not the Rust binary"}


def l0_config(ctx, dev, stream, workload: str, n_loci: int, what: str, reps: int = 10):
    """L0 of one more BASELINE workload in the running process: the batch generated on the device, 3 + `reps` launches with the
    generator's depth as the hint, HIP events around the first kernel and the whole sequence, wall clock around the `reps`."""
    import torch

    from inquistr_amd import synth

    w = synth.WORKLOADS[workload]
    n = n_loci or w.n_loci
    bg_resume()
    shard = synth.DeviceBatch(w, dev, 0, n)
    bg_pause()
    try:
        ctx.set_option("max_reads_hint", w.reads_per_locus)
        for _ in range(3):
            ctx.call_batch_device(shard.c_batch, shard.c_result, stream.cuda_stream)
        torch.cuda.synchronize()
        rc, _ = ctx.status()
        if rc != 0:
            raise RuntimeError(f"device status {rc}")
        ctx.timing_enable(True)
        ctx.timing_reset()
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.call_batch_device(shard.c_batch, shard.c_result, stream.cuda_stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        k_ms, nl = ctx.timing_read(1)
        s_ms, _ = ctx.timing_read(0)
        ctx.timing_enable(False)
        rc, ties = ctx.status()
        if rc != 0:
            raise RuntimeError(f"device status {rc}")
        alg = shard.algorithmic_bytes()
        return {"what": what, "loci": n, "pairs": shard.n_pairs, "cigar_ops": shard.n_ops_total, "unphased": bool(w.unphased),
                "value": n * reps / dt, "unit": "loci/s", "ms_per_step": dt * 1e3 / reps, "avg_kernel_ms": k_ms / nl,
                "avg_launch_sequence_ms": s_ms / nl, "algorithmic_bytes": alg, "achieved_GBps": alg / (k_ms / nl * 1e-3) / 1e9,
                "frac": alg / (k_ms / nl * 1e-3) / 1e9 / HBM_PEAK_GBS, "launches_timed": nl, "n_tie_loci": ties}
    finally:
        del shard
        torch.cuda.empty_cache()
        bg_resume()


def live_traffic(workload: str, per_gpu: int):
    """HBM bytes per launch of the hot kernel from the PMC counters, collected IN THIS RUN: two child processes under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, no trace domains, as /opt/skills/guides/MI355X_MICROARCH.md
    prescribes) run this very script's timed loop for a few steps; FETCH_SIZE is doubled (gfx950 counts a wide coalesced stream at
    half), WRITE_SIZE taken as is, KiB -> bytes, mean over the kernel's dispatches.  None if the profiler is not there or fails."""
    import csv
    import glob
    import shutil
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    out = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="inq_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--steps", "6", "--warmup", "2",
                   "--workload", workload, "--loci-per-gpu", str(per_gpu), "--no-cpu-baseline", "--no-l2", "--no-read-peak", "--no-live-pmc", "--no-l0-configs"]
            # (its own process group, so that a profiler that hangs is ended together with the program it started)
            pr = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), start_new_session=True)
            try:
                pr.communicate(timeout=120)
            except subprocess.TimeoutExpired:
                import signal

                os.killpg(pr.pid, signal.SIGKILL)
                pr.communicate()
                return None
            r = pr
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "locus_call_small" in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                        vals.append(float(row["Counter_Value"]))
            if r.returncode != 0 or not vals:
                return None
            out[counter] = (sum(vals) / len(vals), len(vals))
        except Exception:  # noqa: BLE001
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    fetch_kib, write_kib = out["FETCH_SIZE"][0], out["WRITE_SIZE"][0]
    return {"hbm_bytes_per_launch": 2 * fetch_kib * 1024 + write_kib * 1024, "fetch_size_kib_raw": fetch_kib, "write_size_kib_raw": write_kib,
            "dispatches": [out["FETCH_SIZE"][1], out["WRITE_SIZE"][1]]}


def measured_read_peak():
    """SURVEY 8(d) prices the kernel against the 8 TB/s spec peak AND against what a pure streaming read reaches on the box:
    tools/hbm_read_peak.hip (built by __graft_entry__.build()) reads 2.4 GB with the hot kernel's access shape - one wave per
    24 KB, 16 B per lane, non-temporal - and nothing else; best of 10 launches, HIP events, in a process of its own."""
    exe = os.path.join(ROOT, "inquistr_amd", "lib", "hbm_read_peak")
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe], capture_output=True, text=True, timeout=120).stdout
    except Exception:  # noqa: BLE001  the line stays valid without it
        return None
    best = {}
    for ln in out.splitlines():
        f = ln.split()
        if len(f) >= 4 and f[-1] == "GB/s":
            best[" ".join(f[:-4])] = float(f[-2])
    if "locus-shaped nt" not in best:
        return None
    return {"value": best["locus-shaped nt"], "unit": "GB/s", "all": best,
            "what": "pure read of 2.4 GB, one wave per 24 KB segment, 16 B per lane, non-temporal (the hot kernel's shape), best of 10, measured in this run"}


def _gpu_visible_count() -> int:
    import torch

    return torch.cuda.device_count()  # (counting devices does not initialise the runtime on this image)


def l2_dist(args):
    """`bench.py --gpus N --l2-dist`: the END-TO-END path at N ranks - inquistr_amd.call_dist (one process per GPU, loci cut by BAM
    bytes, rows gathered to rank 0 over RCCL from device memory) over one SEQ-bearing BAM (the l2_seq_large shape: north_star's
    100k-locus / 30x long-read file is --l2-dist-loci 100000), strong scaling.  One step = the whole file through all ranks
    (BAM + BED -> .inq text on rank 0); barrier + device synchronise on both sides of the K timed steps, max over ranks.  The line
    carries every rank's loci, BAM bytes, seconds (rows, gather, span loop GB/s, waiting for its loader, reader threads), the check
    that the .inq is byte-identical to the single-process CLI's, CPU mode B on the same file, and the same file through the
    one-process form (`inquistr call --devices ...`).  What eight links feed from ONE host's page cache is the question this mode
    answers: `host` holds the predicted bound."""
    import statistics
    import tempfile

    import numpy as np
    import torch
    import torch.distributed as dist

    from inquistr_amd import call as hostcall
    from inquistr_amd import call_dist, synth
    from tools import make_synth_bam

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and not (world == 1 and args.gpus == 1):
        raise SystemExit("launch N>1 through torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = args.backend
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    line = l2_dist_core(args, rank, world, local_rank, dev, backend, args.l2_dist_loci, args.steps, args.warmup)
    if rank == 0 and line is not None:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def l2_dist_core(args, rank, world, local_rank, dev, backend, loci, steps, warmup, native=True, cpu_b=True):
    """The measurement of l2_dist() on a process group that exists already (the default `--gpus N` line carries it as its `l2_dist`
    block, N > 1).  Returns the line (rank 0) or None."""
    import statistics
    import tempfile

    import torch
    import torch.distributed as dist

    from inquistr_amd import call as hostcall  # noqa: F401
    from inquistr_amd import call_dist, synth
    from tools import make_synth_bam

    result = None
    wl = synth.WORKLOADS[args.workload]
    threads = args.l2_threads or host_threads()
    # ---- the file: written once by rank 0 (all its cores), kept when --l2-keep names a directory
    tmp = args.l2_keep or (tempfile.mkdtemp(prefix="inq_l2dist_") if rank == 0 else None)
    box = [tmp]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    tmp = box[0]
    loci = min(loci, wl.n_loci)
    prefix = os.path.join(tmp, f"{wl.name}_{loci}_seq{args.l2_level}")
    gen_s = 0.0
    gen_err = [None]
    if rank == 0:
        try:
            os.makedirs(tmp, exist_ok=True)
            if not os.path.exists(prefix + ".bam"):
                t0 = time.perf_counter()
                make_synth_bam.write_native(wl.name, loci, prefix, threads=host_cores_available(), device=dev, seq=True, level=args.l2_level)
                gen_s = time.perf_counter() - t0
            for _ in range(2):  # page cache warm = read TWICE (the second read of a fresh file is the slow one: DESIGN.md 4)
                with open(prefix + ".bam", "rb", buffering=0) as f:
                    while f.read(64 << 20):
                        pass
        except Exception as e:  # noqa: BLE001  (disk full, ...): every rank must learn it, none may wait at the barrier
            gen_err = [f"{type(e).__name__}: {e}"]
    if world > 1:
        dist.broadcast_object_list(gen_err, src=0)
    if gen_err[0]:
        raise RuntimeError("writing the BAM failed on rank 0: " + gen_err[0])
    bam_bytes = os.path.getsize(prefix + ".bam")
    un = wl.unphased
    out_path = os.path.join(tmp, f"dist_rank0_{world}.inq")

    def one_pass(stats):
        with open(out_path if rank == 0 else os.devnull, "w") as f:
            call_dist.genotype_repeats_distributed(prefix + ".bam", None, prefix + ".bed", wl.minlen, wl.support, threads, un, "S", out=f,
                                                   rank=rank, world=world, device=local_rank, frontend="device", stats=stats)

    for _ in range(warmup):
        one_pass({})
    per_step = []
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        st = {}
        one_pass(st)
        per_step.append(st)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt_max = float(tt.item())

    def med(key):
        v = [s[key] for s in per_step if s.get(key) is not None]
        return statistics.median(v) if v else None

    mine = {"rank": rank, "device": local_rank, "loci": per_step[-1].get("loci"), "bam_bytes_read": per_step[-1].get("bam_bytes_this_call"),
            "open_s": med("open_s"), "output_s": med("output_s"), "rows_s": med("rows_s"), "gather_s": med("gather_s"), "span_loop_s": med("span_loop_s"), "span_loop_GBps": med("span_loop_GBps"),
            "wait_loader_s": med("wait_loader_s"), "device_calls_s": med("device_calls_s"), "io_threads": per_step[-1].get("io_threads"),
            "granted_cpus": per_step[-1].get("granted_cpus"), "rows_in": per_step[-1].get("rows_in"), "front": per_step[-1].get("front")}
    allr = [None] * world
    if world > 1:
        dist.all_gather_object(allr, mine)
    else:
        allr = [mine]
    if rank == 0:
        cli = os.path.join(ROOT, "inquistr_amd", "lib", "inquistr")
        ref = os.path.join(ROOT, "oracle", "ref_shaped_call")
        text = open(out_path, "rb").read()
        line = {
            "metric": "loci/sec genotyped (inquiSTR call, BAM + BED -> .inq end to end = L2, one process per GPU)",
            "value": loci * steps / dt_max, "unit": "loci/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": dt_max * 1e3 / steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32/i64",
            "data": "synthetic",
            "config": {"workload": f"{wl.name}: {loci} loci x {wl.reads_per_locus} reads of long-read-shaped records (SEQ + QUAL + ML / MM, HP last), "
                                   f"{bam_bytes / 1e9:.1f} GB of BAM at zlib level {args.l2_level}, " + ("--unphased" if un else "phased (HP)") +
                                   f", inquistr_amd.call_dist at {world} rank(s), {backend} gather of 16 B per locus to rank 0",
                       "loci": loci, "bam_bytes": bam_bytes, "threads_per_rank": threads, "same_device": bool(args.same_device),
                       "bam_gen_s": gen_s},
            "seconds_per_file": dt_max / steps,
            "per_rank": allr,
        }
        reads = [r["bam_bytes_read"] or 0 for r in allr]
        loop = [r["span_loop_s"] for r in allr if r.get("span_loop_s")]
        line["bam_bytes_read_all_ranks"] = int(sum(reads))
        line["bam_bytes_read_over_file"] = sum(reads) / bam_bytes
        if loop:
            # conservative: all compressed bytes of a pass / the WHOLE pass (context creation, planning, rows, gather, text included) - the
            # ranks' span loops are short and need not coincide, so bytes / the longest loop would overstate what the links carried at once
            agg = sum(reads) / 1e9 / (dt_max / steps)
            links = 1 if args.same_device else world
            line["pcie"] = {"bound": "pcie", "achieved": agg, "unit": "GB/s", "peak": PCIE_SPEC_GBS * links, "frac": agg / (PCIE_SPEC_GBS * links),
                            "span_loop_GBps_per_rank": [r.get("span_loop_GBps") for r in allr], "links": links,
                            "what": "compressed bytes all ranks handed to their devices in one pass / the seconds of the whole pass (max over ranks); "
                                    "peak = one Gen5 x16 link per rank" + (" (the ranks share ONE device and link here)" if args.same_device else "") +
                                    "; span_loop_GBps_per_rank = each rank's own bytes / its own span loop"}
        # the host side of N links: what page-cache reads + DMA reads ask of the host's memory per second at the achieved rate
        line["host"] = {"granted_cpus": allr[0].get("granted_cpus"), "reader_threads_per_rank": allr[0].get("io_threads"),
                        "host_memory_traffic_GBps_at_achieved_rate": (3 * line["pcie"]["achieved"]) if loop else None,
                        "what": "every compressed byte is read from the page cache and written to a span buffer by pread (2 x), then read by the "
                                "device's DMA (1 x): 3 x the aggregate link rate is asked of the host's memory (DESIGN.md 5 prices it at N = 2 / 4 / 8)"}
        # ---- byte-identical to the single-process run, and the one-process form on the same devices
        un_flag = ["-u"] if un else []
        base = [cli, "call", prefix + ".bam", "-R", prefix + ".bed", "-t", str(threads), "--sample-name", "S"] + un_flag
        env = dict(os.environ, INQ_FRONTEND="device")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE"):
            env.pop(k, None)
        time.sleep(GPU_REST_S)
        t = time.perf_counter()
        r1 = subprocess.run(base, capture_output=True, env=env, timeout=900)
        t_single = time.perf_counter() - t
        line["single_process_cli"] = {"seconds": t_single, "rc": r1.returncode}
        line["inq_identical_to_single_process"] = bool(r1.returncode == 0 and r1.stdout == text)
        line["speedup_vs_single_process_cli"] = t_single / (dt_max / steps)
        if native and not args.no_l2_dist_native:
            devs = ",".join(str(0 if args.same_device else d) for d in range(world)) if world > 1 else (args.native_devices or "0")
            if "," in devs:
                ts, ok = [], True
                for _ in range(max(1, steps)):
                    time.sleep(GPU_REST_S)
                    t = time.perf_counter()
                    rn = subprocess.run(base + ["--devices", devs], capture_output=True, env=dict(env, INQ_TIMING="1"), timeout=900)
                    ts.append(time.perf_counter() - t)
                    ok = ok and rn.returncode == 0 and rn.stdout == text
                parts = [ln for ln in rn.stderr.decode().splitlines() if ln.startswith("[inq part]")]
                line["native_devices"] = {"devices": devs, "seconds_median": statistics.median(ts), "seconds_all": ts, "loci_per_s": loci / statistics.median(ts),
                                          "inq_identical": bool(ok), "parts": parts,
                                          "what": "`inquistr call --devices " + devs + "`: ONE process, one thread + one device context per device, rows scattered in host "
                                                  "memory (no torch, no collective); whole process start to exit"}
        if cpu_b and not args.no_cpu_baseline:
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref_shaped_call"], stdout=subprocess.DEVNULL)
            cores = host_cores_available()
            t = time.perf_counter()
            rb = subprocess.run([ref, prefix + ".bam", prefix + ".bed", "B", str(cores), str(int(un)), str(wl.minlen), str(wl.support), "S"],
                                capture_output=True, timeout=3000)
            tb = time.perf_counter() - t
            line["cpu_baseline"] = {"value": loci / tb, "unit": "loci/s", "cores": cores, "kind": "port", "seconds": tb,
                                    "inq_identical": bool(rb.returncode == 0 and rb.stdout == text),
                                    "sample": f"the whole file once: oracle/ref_shaped_call mode B (one reader per worker, own BGZF / BAM / BAI reader on zlib) on "
                                              f"{cores} threads - a CPU restatement of the reference's control flow, not the Rust binary"}
            line["speedup_vs_B"] = line["value"] / line["cpu_baseline"]["value"]
        if not args.l2_keep:
            import shutil

            shutil.rmtree(tmp, ignore_errors=True)
        result = line
    if world > 1:
        dist.barrier()
    return result


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="unphased100k", help="phased10k | unphased100k | shard500k | expansion50k")
    ap.add_argument("--loci-per-gpu", type=int, default=0, help="override the per-GPU shard size")
    ap.add_argument("--cpu-sample-loci", type=int, default=20_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-l2", action="store_true", help="skip the end-to-end block of the default N=1 line")
    ap.add_argument("--no-read-peak", action="store_true", help="skip the pure-read ceiling run (a child process; not wanted under a profiler)")
    ap.add_argument("--no-live-pmc", action="store_true", help="roofline.traffic from profiles/pmc_latest.json instead of two rocprofv3 --pmc child runs of this script")
    ap.add_argument("--l2-default-loci", type=int, default=100_000, help="loci of the BAM the default line's l2 block is timed on")
    ap.add_argument("--l2-seq-default-loci", type=int, default=6_000,
                    help="loci of the SEQ / QUAL-bearing BAM the default line's l2_seq block is timed on (30 reads x ~18 KB each per locus)")
    ap.add_argument("--no-l2-seq", action="store_true", help="skip the l2_seq block of the default N=1 line")
    ap.add_argument("--l2-seq-large-loci", type=int, default=-1,
                    help="loci of the large SEQ / QUAL-bearing BAM of the default line's l2_seq_large block (0.32 GB per 1 000 loci; north_star's "
                         "configuration is 100 000 = 32 GB); -1 = the largest size up to 100 000 that disk, page cache and --l2-seq-large-gen-budget admit; 0 skips it")
    ap.add_argument("--l2-seq-large-gen-budget", type=float, default=130.0,
                    help="seconds of all granted cores the large file may cost to write (it is written by a child process beside the earlier blocks, stopped during every timed region)")
    ap.add_argument("--no-l2-phased", action="store_true", help="skip the l2_phased block (config #2's shape as a SEQ-bearing file, phased: the reference's default mode)")
    ap.add_argument("--no-l0-configs", action="store_true", help="skip the per-config L0 figures (phased10k, expansion50k, shard500k / 8)")
    ap.add_argument("--l2-level", type=int, default=6, help="zlib level of the BAMs the l2 blocks are timed on (6 = htslib's default)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="strong: the workload's n_loci are split over the ranks (config #4: --workload shard500k --scaling strong)")
    ap.add_argument("--pmc-summary", default=os.path.join(ROOT, "profiles", "pmc_latest.json"))
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --same-device rehearses the N>1 control flow on a one-GPU box")
    ap.add_argument("--same-device", action="store_true", help="every rank uses cuda:0 (rehearsal only)")
    ap.add_argument("--gather-every", type=int, default=4, help="N>1: steps whose rows travel to rank 0 in one gather")
    ap.add_argument("--l2", action="store_true", help="measure L2 (BAM+BED -> .inq, CLI vs CPU baseline modes) instead of L0")
    ap.add_argument("--l2-loci", type=int, default=20_000)
    ap.add_argument("--l2-threads", type=int, default=0)
    ap.add_argument("--l2-keep", default="", help="directory to keep / reuse the generated BAM in")
    ap.add_argument("--l2-cpu-modes", default="CBA", help="which CPU baseline modes to time (A is slow on large inputs)")
    ap.add_argument("--l2-seq", action="store_true", help="records carry SEQ / QUAL / MM / ML like a real long-read BAM (~30 KB each)")
    ap.add_argument("--l2-dist", action="store_true",
                    help="END-TO-END at N ranks: inquistr_amd.call_dist over one SEQ-bearing BAM, strong scaling (see l2_dist()); --steps = passes over the file")
    ap.add_argument("--l2-dist-loci", type=int, default=20_000, help="loci of that BAM (0.32 GB per 1 000; north_star's configuration: 100000)")
    ap.add_argument("--no-l2-dist-native", action="store_true", help="skip the one-process `inquistr call --devices` run of --l2-dist")
    ap.add_argument("--l2-dist-default-loci", type=int, default=-1,
                    help="N > 1, default line: loci of the SEQ-bearing BAM of its `l2_dist` block (the end-to-end multi-rank measurement beside the L0 one; 0 skips it; "
                         "-1 = the largest size up to north_star's 100 000 that rank 0's granted cores write within --l2-dist-gen-budget seconds and disk / page cache admit)")
    ap.add_argument("--l2-dist-gen-budget", type=float, default=45.0, help="seconds the file of the default N > 1 line's `l2_dist` block may take to write")
    ap.add_argument("--native-devices", default="", help="--l2-dist at one rank: the device list of the one-process run (e.g. 0,0,0,0 on a one-GPU box)")
    args = ap.parse_args()
    if args.l2_dist:
        if args.steps == 30 and args.warmup == 5:  # the defaults are the L0 loop's: a step here is a whole file
            args.steps, args.warmup = 3, 1
        l2_dist(args)
        return
    if args.l2:
        print(json.dumps(l2_end_to_end(args.workload, args.l2_loci, args.l2_threads or host_threads(), 2, args.l2_keep, args.l2_cpu_modes, args.l2_seq)), flush=True)
        return

    import torch
    import torch.distributed as dist

    from inquistr_amd import hipcall, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 through torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import datetime

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=600))
        else:
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=600))

    wl = synth.WORKLOADS[args.workload]
    global BG
    large_plan = None
    if world == 1 and not args.no_l2 and not args.no_l2_seq and args.l2_seq_large_loci != 0:
        # The large SEQ-bearing file (north_star's configuration: 100 000 loci x 30 reads of long-read records = 31 GB at level 6) is
        # written by a child process from now on, beside everything up to its own block.  Its size: the largest up to 100 000 loci
        # that (a) disk and page cache admit and (b) the granted cores deflate within --l2-seq-large-gen-budget seconds at
        # ~55 loci per second and core (measured: 880 loci/s on 16 cores at level 6; level 1 is ~3 x that).
        import shutil
        import tempfile

        want = args.l2_seq_large_loci if args.l2_seq_large_loci > 0 else 100_000
        chosen_by = "--l2-seq-large-loci" if args.l2_seq_large_loci > 0 else "north_star's 100 000 loci"
        if args.l2_seq_large_loci < 0:
            rate = (55.0 if args.l2_level >= 4 else 160.0) * host_cores_available()
            if want > rate * args.l2_seq_large_gen_budget:
                want = int(rate * args.l2_seq_large_gen_budget) // 1000 * 1000
                chosen_by = (f"generation budget: {args.l2_seq_large_gen_budget:.0f} s at ~{rate:.0f} loci/s written "
                             f"(level {args.l2_level}, {host_cores_available()} cores)")
        free = shutil.disk_usage(tempfile.gettempdir()).free
        try:
            avail = int(next(ln for ln in open("/proc/meminfo") if ln.startswith("MemAvailable")).split()[1]) * 1024
        except Exception:  # noqa: BLE001
            avail = 0
        per_locus = 0.33e6 * 1.3
        cap = int(min(free / per_locus, avail / (2 * per_locus))) // 1000 * 1000
        if cap < want:
            want, chosen_by = cap, f"disk / page cache: {free / 1e9:.0f} GB free in {tempfile.gettempdir()}, {avail / 1e9:.0f} GB of memory available"
        want = min(want, wl.n_loci)
        large_plan = {"loci": want, "chosen_by": chosen_by}
        if want >= 5_000:
            try:
                BG = BackgroundGen(wl.name, want, args.l2_level, tempfile.mkdtemp(prefix="inq_l2_large_"))
            except Exception as e:  # noqa: BLE001
                large_plan["error"] = f"{type(e).__name__}: {e}"
    if args.scaling == "strong":
        # total work fixed: the workload's loci are cut into `world` contiguous shards (config #4: 500 000 / N per rank);
        # every rank allocates the largest shard size so the gather buffers are rectangular
        total = args.loci_per_gpu * world if args.loci_per_gpu else wl.n_loci
        per_gpu = (total + world - 1) // world
        lo, hi = min(total, rank * per_gpu), min(total, (rank + 1) * per_gpu)
    elif args.workload == "shard500k":
        # weak form of config #4: the 500k loci are the 8-GPU total; per-GPU shard fixed
        per_gpu = args.loci_per_gpu or wl.n_loci // 8
        lo, hi = rank * per_gpu, (rank + 1) * per_gpu
        total = per_gpu * world
    else:
        per_gpu = args.loci_per_gpu or wl.n_loci
        lo, hi = rank * per_gpu, (rank + 1) * per_gpu
        total = per_gpu * world
    n_mine = hi - lo
    want_l2_dist = world > 1 and not args.no_l2 and args.l2_dist_default_loci != 0 and args.scaling == "weak" and args.workload == "unphased100k"
    l2_dist_loci = args.l2_dist_default_loci
    if want_l2_dist and l2_dist_loci < 0:
        # strong scaling over N links wants a file whose per-rank share is still many spans: the largest up to north_star's 100 000 loci
        # (31 GB) that rank 0's cores deflate within the budget (~55 loci per second and core at level 6) and the box can hold twice
        import shutil
        import tempfile

        rate = (55.0 if args.l2_level >= 4 else 160.0) * host_cores_available()
        l2_dist_loci = min(100_000, max(8_000, int(rate * args.l2_dist_gen_budget) // 1000 * 1000))
        try:
            avail = int(next(ln for ln in open("/proc/meminfo") if ln.startswith("MemAvailable")).split()[1]) * 1024
            free = shutil.disk_usage(tempfile.gettempdir()).free
            l2_dist_loci = max(2_000, min(l2_dist_loci, int(min(free / 0.43e6, avail / 0.86e6)) // 1000 * 1000))
        except Exception:  # noqa: BLE001
            pass
        pick = [l2_dist_loci]
        dist.broadcast_object_list(pick, src=0)  # (every rank must name the same file)
        l2_dist_loci = pick[0]
    final_line = None

    ctx = hipcall.Context(local_rank)
    ctx.set_option("max_reads_hint", wl.reads_per_locus)  # the generator's fixed depth: no deep-locus launches needed
    shard = synth.DeviceBatch(wl, dev, lo, hi)
    # results: two rotating groups of G steps, each [G, 2, n] (row 0 = H1, row 1 = H2).  The rows of a whole
    # group go to rank 0 in ONE gather (fewer, larger collectives) that overlaps the next group's kernels.
    G = max(1, args.gather_every) if world > 1 else 1
    outs = [torch.empty(G, 2, per_gpu, dtype=torch.float64, device=dev) for _ in range(2)]
    gdev = dev if args.backend == "nccl" else torch.device("cpu")
    gathered = [torch.empty(world, G, 2, per_gpu, dtype=torch.float64, device=gdev) for _ in range(2)] if (world > 1 and rank == 0) else None
    stage = [torch.empty(G, 2, per_gpu, dtype=torch.float64).pin_memory() for _ in range(2)] if (world > 1 and args.backend == "gloo") else None
    comm_stream = torch.cuda.Stream(device=dev) if world > 1 else None
    # an explicit stream: torch's default stream has handle 0, which the C ABI reads as "the ctx's own stream" - the kernels would
    # then run on a stream the events below know nothing about, and the gather of a group could read rows still being written
    main_stream = torch.cuda.Stream(device=dev)
    assert main_stream.cuda_stream != 0
    events = [torch.cuda.Event() for _ in range(2)]

    from inquistr_amd.batch import InqResultC

    def result_for(buf):
        r = InqResultC()
        r.phase1, r.phase2 = buf[0].data_ptr(), buf[1].data_ptr()
        r.pair_call = r.pair_bits = None
        return r

    results = [[result_for(outs[b][g]) for g in range(G)] for b in range(2)]
    pending = []
    # ONE stated collective: gather to rank 0 (ProcessGroupNCCL::gather = grouped ncclSend / ncclRecv over xGMI; 16 B per locus).
    # No fallback: if this RCCL build cannot do it the run fails loudly (tests/test_gpu_parity.py::test_nccl_gather_is_available
    # exercises the call on the GPU box).
    def send_group(b):
        ev = events[b]
        ev.record(main_stream)
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(ev)
            glist = list(gathered[b].unbind(0)) if rank == 0 else None
            if args.backend == "nccl":
                pending.append(dist.gather(outs[b], glist, dst=0, async_op=True))
            else:  # rehearsal: stage through pinned host memory, gather on gloo
                stage[b].copy_(outs[b], non_blocking=True)
                comm_stream.synchronize()
                pending.append(dist.gather(stage[b], glist, dst=0, async_op=True))

    state = {"n": 0}  # steps issued since the last drain

    def step(_i):
        n = state["n"]
        b, g = (n // G) & 1, n % G
        if world > 1 and g == 0 and len(pending) >= 2:
            # the group buffer b is free again once its gather is done.  An NCCL work's wait() orders the CURRENT stream behind
            # the collective (it does not block the host), so it has to be called with the compute stream current.
            with torch.cuda.stream(main_stream):
                pending.pop(0).wait()
        ctx.call_batch_device(shard.c_batch, results[b][g], main_stream.cuda_stream)
        state["n"] = n + 1
        if world > 1 and g == G - 1:
            send_group(b)

    def flush():
        n = state["n"]
        if world > 1 and n % G != 0:  # a partly filled group still has to reach rank 0
            send_group((n // G) & 1)
        state["last"] = ((n - 1) // G) & 1, (n - 1) % G
        state["n"] = 0

    def drain():
        flush()
        with torch.cuda.stream(main_stream):
            while pending:
                pending.pop(0).wait()
        torch.cuda.synchronize()

    bg_pause()  # the timed loop has the host to itself
    for i in range(args.warmup):
        step(i)
    drain()
    rc, _ = ctx.status()
    if rc != 0:
        raise SystemExit(f"device status {rc}: {hipcall.strerror(rc)}")

    ctx.timing_enable(True)
    ctx.timing_reset()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    main_stream.synchronize()  # this rank's kernels are through; what follows is the tail of the gathers nothing overlaps any more
    dt_compute = time.perf_counter() - t0
    drain()
    dt_own = time.perf_counter() - t0  # this rank's steps + its gathers, before waiting for the other ranks
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kern_ms, launches = ctx.timing_read(1)  # HIP events on the launch stream around the CIGAR-walk kernel
    seq_ms, _ = ctx.timing_read(0)
    ctx.timing_enable(False)
    rc, ties = ctx.status()
    if rc != 0:
        raise SystemExit(f"device status {rc}: {hipcall.strerror(rc)}")
    # The same launch WITHOUT the caller's depth hint: an external caller of inq_call_batch_device that does not know its deepest
    # locus gets two more launches behind the first kernel (locus_call_mid_walk and the persistent locus_call_tail, which find
    # empty work lists and leave; round 4: ~38 launches, 0.15 ms).  Same batch, same stream, HIP events around the whole sequence.
    no_hint = None
    if world == 1:
        ctx.set_option("max_reads_hint", 0)
        for i in range(3):
            ctx.call_batch_device(shard.c_batch, results[0][0], main_stream.cuda_stream)
        torch.cuda.synchronize()
        ctx.timing_enable(True)
        ctx.timing_reset()
        n_nh = max(5, min(args.steps, 20))
        for i in range(n_nh):
            ctx.call_batch_device(shard.c_batch, results[0][0], main_stream.cuda_stream)
        torch.cuda.synchronize()
        nh_seq, nh_n = ctx.timing_read(0)
        nh_k, _ = ctx.timing_read(1)
        ctx.timing_enable(False)
        rc2, _t = ctx.status()
        if rc2 != 0:
            raise SystemExit(f"device status {rc2} without the depth hint: {hipcall.strerror(rc2)}")
        ctx.set_option("max_reads_hint", wl.reads_per_locus)
        no_hint = {"avg_launch_sequence_ms": nh_seq / nh_n, "avg_kernel_ms": nh_k / nh_n, "launches_timed": nh_n}
    bg_resume()

    t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t.item())
    # every rank's own clock, so that a scaling run explains itself: its steps, the tail of the gathers behind them (exposed: not
    # hidden behind kernels), its wait at the closing barrier
    mine = torch.tensor([dt_compute, dt_own, dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
    if world > 1:
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
    else:
        allr = [mine]
    per_rank = [{"rank": r, "ms_per_step": float(x[1]) * 1e3 / args.steps, "kernels_ms_per_step": float(x[0]) * 1e3 / args.steps,
                 "gather_exposed_ms_total": (float(x[1]) - float(x[0])) * 1e3, "barrier_wait_ms": (float(x[2]) - float(x[1])) * 1e3}
                for r, x in enumerate(allr)]

    if rank == 0 and world > 1:
        # the last gathered group must hold rank 0's own rows in slot 0
        lb, lg = state["last"]
        src = gathered[lb]
        own = outs[lb][lg].to(src.device)
        g0 = src[0][lg]
        if not bool(((own == g0) | (own.isnan() & g0.isnan())).all()):
            raise SystemExit("gathered rows differ from the local result")
    if rank == 0:
        total_loci = total
        alg_bytes = shard.algorithmic_bytes()
        avg_kernel_s = kern_ms / 1e3 / max(1, launches)
        achieved = alg_bytes / avg_kernel_s / 1e9
        # PMC counters cannot be read from inside this process: `traffic` is the figure STORED by the last rocprofv3 --pmc run of
        # this very command and workload (profiles/pmc_latest.json, written by tools/summarize_profile.py), labelled as such
        traffic = None
        pmc_note = None
        if os.path.exists(args.pmc_summary):
            try:
                pmc = json.load(open(args.pmc_summary))
                if pmc.get("workload") == wl.name and pmc.get("loci_per_gpu") == per_gpu:
                    traffic = pmc.get("hbm_bytes_per_launch")
                    pmc_note = {"kind": "stored", "source": pmc.get("source"), "commit": pmc.get("commit"), "profile": pmc.get("profile")}
            except Exception:
                traffic = None
        if world == 1 and not args.no_live_pmc:
            lt = live_traffic(wl.name, per_gpu)
            if lt:
                traffic = lt["hbm_bytes_per_launch"]
                pmc_note = {"kind": "live", "how": "two child runs of this script under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), "
                            "2 x FETCH_SIZE (gfx950 wide-stream correction) + WRITE_SIZE, KiB -> bytes, mean over the kernel's dispatches", **lt}
        line = {
            "metric": "loci/sec genotyped (inquiSTR call hot path, device-resident batch = L0)",
            "value": total_loci * args.steps / dt_max,
            "unit": "loci/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u32/i64",
            "data": "synthetic",
            "config": {
                "workload": f"{wl.name}: {per_gpu} loci/GPU x {wl.reads_per_locus} reads x ~{shard.n_ops_total / max(1, shard.n_pairs):.0f} CIGAR ops, "
                + ("--unphased" if wl.unphased else "phased (HP)")
                + ", device-resident SoA (L0)",
                "loci_per_gpu": per_gpu,
                "pairs_per_gpu": shard.n_pairs,
                "cigar_ops_per_gpu": shard.n_ops_total,
                "minlen": wl.minlen,
                "support": wl.support,
                "sharding": f"loci x {world} ranks ({args.backend} gather), "
                f"rows of {G} steps per collective to rank 0 (16 B/locus), overlapped" if world > 1 else "single GPU",
                "total_loci": total_loci,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "locus_call_small",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes,
                "avg_kernel_ms": avg_kernel_s * 1e3,
                "avg_launch_sequence_ms": seq_ms / max(1, launches),
                "launches_timed": launches,
                "traffic_source": pmc_note,
            },
            "n_tie_loci": ties,
        }
        if world > 1:
            line["per_rank"] = per_rank
            line["gather"] = {"collective": f"dist.gather ({args.backend})", "bytes_per_rank_per_collective": int(G * 2 * per_gpu * 8),
                              "collectives": (args.steps + G - 1) // G,
                              "exposed_ms_total_max_over_ranks": max(p["gather_exposed_ms_total"] for p in per_rank)}
        if no_hint:
            # what an external caller of inq_call_batch_device pays when it cannot promise a depth
            line["roofline"]["avg_launch_sequence_ms_no_hint"] = no_hint["avg_launch_sequence_ms"]
            line["roofline"]["frac_no_hint"] = alg_bytes / (no_hint["avg_launch_sequence_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            line["roofline"]["no_hint"] = dict(no_hint, launches_per_sequence=3, over_hinted_sequence=no_hint["avg_launch_sequence_ms"] / (seq_ms / max(1, launches)),
                                               what="the same launches without max_reads_hint: locus_call_small + locus_call_mid_walk + the persistent locus_call_tail "
                                                    "(both find empty work lists and leave); frac_no_hint = algorithmic bytes / that whole sequence / 8 TB/s")
        if world == 1 and not args.no_read_peak:
            pk = measured_read_peak()
            if pk:
                line["roofline"]["peak_measured"] = pk
                line["roofline"]["frac_of_measured"] = achieved / pk["value"]
        if world == 1 and not args.no_l0_configs:
            # loci/s and roofline fraction per workload, a few timed launches each in this same process (the headline
            # workload is the line itself); config #4 as ONE of its eight shards (62 500 loci: what one GPU of the node holds)
            line["l0_configs"] = {}
            for name, n_cfg, what in (("phased10k", 0, "config #2"), ("expansion50k", 0, "config #5"), ("shard500k", 500_000 // 8, "config #4, one of 8 shards")):
                if name == wl.name:
                    continue
                try:
                    line["l0_configs"][name] = l0_config(ctx, dev, main_stream, name, n_cfg, what)
                except Exception as e:  # noqa: BLE001
                    line["l0_configs"][name] = {"error": f"{type(e).__name__}: {e}"}
            ctx.set_option("max_reads_hint", wl.reads_per_locus)
        if world == 1 and not args.no_cpu_baseline:
            bg_pause()
            n_s = min(args.cpu_sample_loci, n_mine)
            want, line["cpu_baseline"] = cpu_baseline(wl, n_s)
            bg_resume()
            # the rows the timed steps left on the device for those loci against the oracle's, bit for bit
            lb, lg = state["last"]
            got = outs[lb][lg][:, :n_s].cpu().numpy()
            import numpy as np

            for k, w in ((0, want.phase1), (1, want.phase2)):
                same = (got[k] == w) | (np.isnan(got[k]) & np.isnan(w))
                if not bool(same.all()):
                    raise SystemExit(f"parity: H{k + 1} of locus {int(np.argmin(same))} differs from the CPU oracle")
            line["parity_checked_loci"] = n_s
            line["parity"] = "rows of the cpu_baseline sample produced by the timed steps == CPU oracle rows (bit-exact, NaN == NaN)"
        more_to_come = world == 1 and not args.no_l2
        if more_to_come:
            # The line as far as it exists - metric, roofline, cpu_baseline, the per-config L0 figures - goes out NOW: the end-to-end blocks
            # below write and read tens of GB, and a box with a slow disk must not lose the headline with them.  The full line (a
            # superset of this one) is printed last.
            print(json.dumps(dict(line, partial="L0 + roofline + cpu_baseline; the full line with the end-to-end blocks follows")), flush=True)
        if world == 1 and not args.no_l2:
            # The end-to-end blocks.  Files are written at zlib level 6 (htslib's default for BAM output).
            # At most three GPU processes at any time: this one, and either a CLI run or a server with its one caller.
            floor = None
            h2d = None
            try:
                floor = startup_floor()
                bg_pause()
                h2d = h2d_copy_peak(dev)
            except Exception as e:  # noqa: BLE001
                line["l2_probe_error"] = f"{type(e).__name__}: {e}"
            if floor:
                line["startup_floor"] = floor
            if h2d:
                line["h2d_copy_peak"] = h2d
            try:
                bg_pause()
                line["l1"] = l1_block(wl, local_rank)
            except Exception as e:  # noqa: BLE001
                line["l1"] = {"error": f"{type(e).__name__}: {e}"}
            bg_resume()
            lvl = args.l2_level
            try:
                line["l2"] = l2_block(wl.name, min(args.l2_default_loci, wl.n_loci), host_threads(), dev, level=lvl, floor=floor, h2d=h2d, cohort=False,
                                      a_loci=4_000, b2b_runs=2)
            except Exception as e:  # noqa: BLE001  the L0 line above stays valid without it
                line["l2"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_l2 and not args.no_l2_seq:
            try:  # the same comparison on records shaped like a real long-read BAM (SEQ, QUAL, ML / MM, HP last)
                line["l2_seq"] = l2_block(wl.name, min(args.l2_seq_default_loci, wl.n_loci), host_threads(), dev, a_loci=2_000, c_loci=300, seq=True,
                                          level=lvl, cohort=False, floor=floor, h2d=h2d, b2b_runs=2)
            except Exception as e:  # noqa: BLE001
                line["l2_seq"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_l2 and not args.no_l2_seq and not args.no_l2_phased:
            try:  # PHASED - the reference's default mode (-u is opt-in, src/main.rs:55) - on config #2's shape as a SEQ-bearing file
                wp = synth.WORKLOADS["phased10k"]
                line["l2_phased"] = l2_block(wp.name, wp.n_loci, host_threads(), dev, reps=5, seq=True, level=lvl, lean=True, cohort=False,
                                             floor=floor, h2d=h2d, trace_runs=3, b2b_runs=2)
            except Exception as e:  # noqa: BLE001
                line["l2_phased"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and large_plan is not None:
            # ... and on a file large enough for the process's fixed costs (the HIP runtime's start-up, 0.15 - 0.2 s) not to be the
            # measurement: north_star's own configuration is 100 000 loci x 30 reads of long-read records = 31 GB of BAM.  The file
            # has been in the making since the first second of this run (BackgroundGen); what is left of it is written now.
            if BG is None:
                line["l2_seq_large"] = {"skipped": f"no room or time for a large file ({large_plan['chosen_by']})", **({"error": large_plan["error"]} if "error" in large_plan else {})}
            else:
                try:
                    t_w = time.perf_counter()
                    ok = BG.wait(timeout=max(60.0, 3.0 * args.l2_seq_large_gen_budget))
                    waited = time.perf_counter() - t_w
                    if not ok:
                        raise RuntimeError("the large file's writer failed: " + open(BG.prefix + ".gen.log").read()[-300:])
                    gen = BG
                    BG = None  # (its process is gone: nothing to stop any more)
                    line["l2_seq_large"] = l2_block(wl.name, large_plan["loci"], host_threads(), dev, reps=5, seq=True, lean=True, level=lvl,
                                                    cohort=False, floor=floor, h2d=h2d, trace_runs=4, ready_prefix=gen.prefix, ready_gen_s=gen.seconds,
                                                    b2b_runs=2)
                    line["l2_seq_large"]["size_chosen_by"] = large_plan["chosen_by"]
                    line["l2_seq_large"]["written"] = {"wall_s_start_to_done": gen.seconds, "of_which_stopped_s": gen.paused_s,
                                                       "waited_for_it_at_the_end_s": waited,
                                                       "how": "a niced child process from the first second of this run, stopped during every timed region"}
                except Exception as e:  # noqa: BLE001
                    line["l2_seq_large"] = {"error": f"{type(e).__name__}: {e}"}
                    if BG is not None:
                        BG.abort()
                        BG = None
        final_line = line
        if not (world > 1 and want_l2_dist):
            print(json.dumps(line), flush=True)
    if BG is not None:
        BG.abort()
    if world > 1 and want_l2_dist:
        # N > 1: the END-TO-END measurement beside the L0 one, in the same run - call_dist over one SEQ-bearing BAM, strong scaling
        # (l2_dist_core) - so that the first run on a node of several GPUs shows what their links ask of ONE host and not
        # only that device-resident kernels scale.  The L0 line goes out first; the full line, a superset, last.
        if rank == 0:
            print(json.dumps(dict(final_line, partial="L0 weak scaling; the full line with the end-to-end `l2_dist` block follows")), flush=True)
        block = None
        try:
            block = l2_dist_core(args, rank, world, local_rank, dev, args.backend, l2_dist_loci, 2, 1)
        except Exception as e:  # noqa: BLE001
            block = {"error": f"{type(e).__name__}: {e}"}
        if rank == 0:
            final_line["l2_dist"] = block
            print(json.dumps(final_line), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
