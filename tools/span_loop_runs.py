#!/usr/bin/env python3
"""Runs `inquistr call` on one BAM several times per environment variant and prints, per run: whole-process wall time, the span
loop's own time and rate (CLI's clock, INQ_TIMING=1), time waiting for the loader / in device calls, the CPU time the cgroup burned
and its throttle count.  usage: span_loop_runs.py PREFIX RUNS [-t N] [--unphased] VAR=val,VAR=val ...   ('-' = no variables)"""
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "inquistr_amd", "lib", "inquistr")


def cpu_stat():
    d = {}
    try:
        for ln in open("/sys/fs/cgroup/cpu.stat"):
            k, v = ln.split()
            d[k] = int(v)
    except OSError:
        pass
    return d


def vmstat(keys=("pgactivate", "pgdeactivate", "numa_hint_faults", "thp_fault_alloc", "thp_fault_fallback")):
    d = {}
    try:
        for ln in open("/proc/vmstat"):
            k, v = ln.split()
            if k in keys:
                d[k] = int(v)
    except OSError:
        pass
    return d


def main():
    prefix, runs = sys.argv[1], int(sys.argv[2])
    rest = sys.argv[3:]
    threads = "16"
    if "-t" in rest:
        i = rest.index("-t")
        threads = rest[i + 1]
        del rest[i : i + 2]
    un = []
    if "--unphased" in rest:
        rest.remove("--unphased")
        un = ["-u"]
    keep_dir = None  # --keep-slow DIR: stderr (INQ_TIMING=2) of every run whose span loop is below 30 GB/s, and of the first normal one
    if "--keep-slow" in rest:
        i = rest.index("--keep-slow")
        keep_dir = rest[i + 1]
        del rest[i : i + 2]
        os.makedirs(keep_dir, exist_ok=True)
    variants = rest or ["-"]
    ref = None
    kept_normal = False
    for v in variants:
        env = dict(os.environ, INQ_FRONTEND="device", INQ_TIMING="2" if (keep_dir or os.environ.get("SHOW_CALLS")) else "1")
        if v != "-":
            for kv in v.split(","):
                k, val = kv.split("=", 1)
                env[k] = val
        loops = []
        for i in range(runs):
            if os.environ.get("PAUSE_S"):  # let the previous process's teardown (the driver wipes the VRAM it held) finish first
                time.sleep(float(os.environ["PAUSE_S"]))
            a = cpu_stat()
            va = vmstat()
            t0 = time.perf_counter()
            r = subprocess.run([CLI, "call", prefix + ".bam", "-R", prefix + ".bed", "-t", threads, "--sample-name", "S"] + un, capture_output=True, env=env)
            dt = time.perf_counter() - t0
            b = cpu_stat()
            vb = vmstat()
            err = r.stderr.decode()
            if r.returncode != 0:
                print(f"{v}: run {i} FAILED rc {r.returncode}: {err[-300:]}")
                continue
            if ref is None:
                ref = r.stdout
            same = "same" if r.stdout == ref else "DIFFERENT OUTPUT"
            m = re.search(r"span loop: (\d+) spans, ([\d.]+) MB compressed, ([\d.]+) s .* = ([\d.]+) GB/s", err)
            w = re.search(r"waiting for the loader ([\d.]+)s, device calls ([\d.]+)s", err)
            loop_s, rate = (float(m.group(3)), float(m.group(4))) if m else (0.0, 0.0)
            # the locus kernels of every flush: loci, MB of CIGARs, kernel ms -> algorithmic TB/s (4 B per op + 20 B per pair at 30 pairs per locus + 32 B per locus)
            calls = ["%d loci %.3f ms = %.2f TB/s" % (int(a_), float(c_), (float(b_) * 1e6 + int(a_) * (30 * 20 + 32)) / (float(c_) * 1e-3) / 1e12)
                     for a_, b_, c_ in re.findall(r"\[inq call\].*? (\d+) loci, ([\d.]+) MB of CIGARs: locus kernels ([\d.]+) ms", err)]
            loops.append(rate)
            if keep_dir and (rate < 30.0 or not kept_normal):
                tag = "slow" if rate < 30.0 else "normal"
                kept_normal = kept_normal or rate >= 30.0
                open(os.path.join(keep_dir, f"{tag}_{v.replace('=', '').replace(',', '_')}_{i}.err", ), "w").write(err)
            print(f"{v:40s} run {i}: wall {dt:6.3f} s | span loop {loop_s:7.4f} s = {rate:6.2f} GB/s | loader wait {w.group(1) if w else '?'} s, device {w.group(2) if w else '?'} s | "
                  f"cpu {(b.get('usage_usec', 0) - a.get('usage_usec', 0)) / 1e6:6.2f} s, throttled +{b.get('nr_throttled', 0) - a.get('nr_throttled', 0)} | pgactivate +{vb.get('pgactivate', 0) - va.get('pgactivate', 0)}"
                  f" thp_fallback +{vb.get('thp_fault_fallback', 0) - va.get('thp_fault_fallback', 0)} | {same}"
                  + (" | locus kernels: " + "; ".join(calls) if calls and os.environ.get("SHOW_CALLS") else ""), flush=True)
        if loops:
            s = sorted(loops)
            print(f"{v:40s} span loop GB/s: min {s[0]:.2f} median {s[len(s) // 2]:.2f} max {s[-1]:.2f}  ({len(loops)} runs)", flush=True)


if __name__ == "__main__":
    main()
