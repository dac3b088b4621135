"""`inquiSTR call` over several GPUs of one node: one process per GPU (torch.distributed; backend
`nccl` = RCCL on the GPUs, `gloo` for CPU rehearsal), loci partitioned by rank, no data-path
collective; the only exchange is the gather of the per-shard rows to rank 0, which writes the `.inq`.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        -m inquistr_amd.call_dist sample.bam -R loci.bed -t 8 > sample.inq

Each rank runs the C++ driver on its own contiguous slice of the (contig, start)-sorted targets: it reads
only the part of the BAM that slice needs and inflates, scans and calls it on its own GPU (device front
end), or sweeps it on the host for small inputs; the reference's counterpart is the
rayon loop over loci (src/call.rs:115-136), which shares nothing but the output Vec.

Where the rows are gathered from: on `nccl` (= RCCL) every rank's rows STAY in device memory (inq_run_rows_device:
each flush of the locus kernels scatters its rows into the rank's [2][width] device buffer) and the one collective
reads them there - no host -> device copy to satisfy the backend; rank 0 copies the gathered block down once.  On
`gloo` (CPU rehearsal) the rows come back as host arrays and are gathered as they are.  The one-process form
(`inquistr call --devices 0,1,...`, host/multi_device.cc) needs neither: its gather is a scatter in host memory.
"""
from __future__ import annotations

import argparse
import os
import sys
import tempfile
from typing import Callable, List, Optional

import numpy as np

from . import call as hostcall


class _DeviceArray:
    """A device allocation of the C++ library as something torch.as_tensor wraps without copying."""

    def __init__(self, ptr: int, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


_sessions = {}


def _process_session(device: int):
    """One call.Session per device for the life of the process (closed at exit): a rank's device context, its span buffers and
    its parsed BED outlive the file."""
    sess = _sessions.get(device)
    if sess is None:
        import atexit

        sess = _sessions[device] = hostcall.Session(device)
        atexit.register(sess.close)
    return sess


def _drop_process_session(device: int) -> None:
    sess = _sessions.pop(device, None)
    if sess is not None:
        try:
            sess.close()
        except Exception:  # noqa: BLE001
            pass


def _exchange_status(status: int, message: str, rank: int, world: int, group=None):
    """Every rank learns whether any rank failed (and rank 0 the failing rank's message) BEFORE the row gather, so that a
    data-dependent failure on one rank (a record the reference panics on, a HIP error, out of memory) ends the run on all
    ranks with the same exit status instead of leaving the others waiting in the gather."""
    import torch.distributed as dist

    if world == 1:
        return status, message, rank
    got = [None] * world
    dist.all_gather_object(got, (int(status), str(message)), group=group)
    for r, (st, msg) in enumerate(got):
        if st != 0:
            return st, msg, r
    return 0, "", -1


def genotype_repeats_distributed(bamp: str, region: Optional[str], region_file: Optional[str], minlen: int = 5,
                                 support: int = 3, threads: int = 1, unphased: bool = False,
                                 sample_name: Optional[str] = None, out=None, rank: int = 0, world: int = 1,
                                 device: int = 0, compute: Optional[Callable] = None, group=None,
                                 frontend: Optional[str] = None, stats: Optional[dict] = None, rows: Optional[str] = None,
                                 session=None) -> None:
    """Same arguments as call.genotype_repeats plus (rank, world).  Rank 0 writes header + rows.
    Raises CallError (same status on every rank) if any rank fails.  stats (rank 0): seconds of the output stage.
    rows: "device" / "host" = where this rank's rows wait for the gather; None = device memory when the backend is nccl.
    session: the call.Session this rank's device work runs on; None = one per process and device, opened at the first call and
    kept (a resident rank calls file after file on one device context)."""
    import torch
    import torch.distributed as dist

    # ---- the work split: EVERY rank opens the run (BAM header, index, BED: ~25 ms for 100 000 targets) at the same time and cuts the
    # targets itself - contiguous slices in file order, about the same amount of BAM per rank; the cut is a pure
    # function of the index and the target list, so all ranks hold the same plan without a broadcast (round 4: rank 0 opened, cut and
    # broadcast 100 000 indices while the others waited, then they opened - twice the fixed cost in front of every rank's spans).  One
    # small exchange makes sure of it: status, message and a digest of the plan.  A resident rank (a worker that calls file after
    # file) runs on ONE session per device: the device context, a GB of touched span buffers and the parsed BED are the process's,
    # not the file's (inq_session_run_open).
    import time
    import zlib

    if rows not in (None, "device", "host"):
        raise ValueError("rows: 'device', 'host' or None")
    if rows == "device" and world > 1 and dist.get_backend(group) != "nccl":
        raise ValueError("rows='device' needs the nccl backend")
    t_open = time.perf_counter()
    st, msg = 0, ""
    run = None
    order, cuts, digest = None, None, 0
    try:
        sess = _process_session(device) if (compute is None and session is None) else session
        run = hostcall.Run(bamp, region, region_file, minlen, support, threads, unphased, sample_name, device=device, frontend=frontend,
                           session=sess if compute is None else None)
        order, cuts = run.partition(world)
        digest = zlib.crc32(cuts.tobytes(), zlib.crc32(order.tobytes()))
    except hostcall.CallError as e:
        st, msg = e.status, e.message
    except Exception as e:  # noqa: BLE001  (library missing, out of memory, ...): the other ranks must not wait for this one
        st, msg = 1, f"{type(e).__name__}: {e}"
    if world > 1:
        got = [None] * world
        dist.all_gather_object(got, (int(st), str(msg), int(digest)), group=group)
        for r, (s_r, m_r, _d) in enumerate(got):
            if s_r != 0:
                st, msg = s_r, m_r
                break
        else:
            if any(d != got[0][2] for _s, _m, d in got):
                st, msg = 1, "the ranks cut the targets differently (different files, or different versions of the host library?)"
    if st != 0:
        if run is not None:
            run.close()
        raise hostcall.CallError(st, msg)
    if stats is not None:
        stats["open_s"] = time.perf_counter() - t_open  # header + index + BED + the cut (+ the exchange's wait for the slowest rank)
    n = len(order)
    lo, hi = int(cuts[rank]), int(cuts[rank + 1])
    mine = order[lo:hi]
    width = max(max(int(cuts[r + 1] - cuts[r]) for r in range(world)), 1)
    # rows stay in device memory for the RCCL gather (world 1 + rows="device": the same path without a collective, for tests)
    on_device = compute is None and (rows == "device" or (rows is None and world > 1 and dist.get_backend(group) == "nccl"))
    # ---- this rank's rows
    t_rows = time.perf_counter()
    p1 = np.full(len(mine), np.nan)
    p2 = np.full(len(mine), np.nan)
    dev_buf = None
    st, msg = 0, ""
    try:
        if compute is None and on_device:
            # the device row buffer lives in the run until the gather is over
            d1, _d2 = run.rows_device(mine, width)
            dev_buf = torch.as_tensor(_DeviceArray(d1, (2, width)), device=torch.device("cuda", device))
        elif len(mine) and compute is None:
            # the product path: the C++ driver on this rank's share (inq_genotype_repeats_rows), which picks the device front end
            # (inflate + record scan + join on this rank's GPU) or the host sweep by the amount of BAM; rows come back as f64
            p1, p2 = run.rows(mine)
        elif len(mine):  # tests: per-batch compute supplied by the caller (the oracle, on CPU-only machines)
            fe_all = hostcall.FrontEnd(bamp, region=region, region_file=region_file)
            targets = fe_all.targets()
            fe_all.close()
            with tempfile.NamedTemporaryFile("w", suffix=".bed", delete=False) as f:
                for i in mine:
                    f.write(f"{targets[i][0]}\t{targets[i][1]}\t{targets[i][2]}\n")
                sub_bed = f.name
            try:
                fe = hostcall.FrontEnd(bamp, region_file=sub_bed, minlen=minlen, support=support, threads=threads,
                                       unphased=unphased, sample_name=sample_name)
                for batch, idx in fe.batches():
                    a, b = compute(batch)
                    p1[idx], p2[idx] = a, b
                fe.close()
            finally:
                os.unlink(sub_bed)
    except hostcall.CallError as e:
        st, msg = e.status, e.message
    except Exception as e:  # noqa: BLE001  anything else (HIP runtime, memory) is an error exit on every rank too
        st, msg = 1, f"{type(e).__name__}: {e}"
    if stats is not None:  # every rank: its share, its time, the BAM bytes its device front end was handed
        stats["rank"], stats["world"], stats["loci"] = rank, world, len(mine)
        stats["rows_s"] = time.perf_counter() - t_rows
        stats["rows_in"] = "device memory" if on_device else "host memory"
        try:
            L = hostcall.load()
            stats["bam_bytes_read"] = int(L.inq_host_span_bytes_read())
            stats["granted_cpus"] = int(L.inq_host_granted_cpus())
            last = hostcall.last_call_stats()  # this rank's call: its span loop, its loader waits, its reader pool
            stats.update({k: last[k] for k in ("spans", "span_loop_s", "wait_loader_s", "device_calls_s", "io_threads")})
            stats["front"] = {1: "host", 2: "device"}.get(last["front"], "-")
            stats["bam_bytes_this_call"] = int(last["bam_bytes_read"])
            stats["span_loop_GBps"] = last["bam_bytes_read"] / 1e9 / last["span_loop_s"] if last["span_loop_s"] > 0 else None
        except Exception:  # noqa: BLE001
            stats["bam_bytes_read"] = None
    own_failure = st
    st, msg, bad_rank = _exchange_status(st, msg, rank, world, group)
    if st != 0:
        run.close()
        if own_failure == 1 and compute is None and session is None:
            # an error exit of THIS rank's device work (HIP, memory, no device - not one of the reference's panics, which a session
            # survives): the process's session is given up, the next call makes a new context
            _drop_process_session(device)
        raise hostcall.CallError(st, f"rank {bad_rank}: {msg}" if world > 1 else msg)
    # ---- the one exchange of the path: 2 x f64 per locus to rank 0 (at world size 1 inside an initialised nccl group the same
    # collective runs over the device buffer: the path a one-GPU box can exercise)
    if world > 1 or (on_device and dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl"):
        if on_device:
            buf = dev_buf  # [2][width] in this rank's device memory, written by its flushes
        else:
            buf = torch.full((2, width), float("nan"), dtype=torch.float64)
            buf[0, : len(mine)] = torch.from_numpy(p1)
            buf[1, : len(mine)] = torch.from_numpy(p2)
        bufs = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
        t_g = time.perf_counter()
        dist.gather(buf, bufs, dst=0, group=group)
        if on_device:
            torch.cuda.synchronize(device)  # (the collective is enqueued; the run's buffer must outlive it)
        if stats is not None:
            stats["gather_s"] = time.perf_counter() - t_g  # exposed: nothing overlaps it in a one-file call (16 B per locus)
        if rank != 0:
            run.close()
            return
        full1, full2 = np.full(n, np.nan), np.full(n, np.nan)
        got = torch.stack(bufs).cpu().numpy() if on_device else [b.numpy() for b in bufs]  # rank 0: ONE copy down
        for r in range(world):
            sl = order[int(cuts[r]) : int(cuts[r + 1])]
            full1[sl] = got[r][0, : len(sl)]
            full2[sl] = got[r][1, : len(sl)]
    else:
        full1, full2 = np.full(n, np.nan), np.full(n, np.nan)
        if on_device:
            got = dev_buf.cpu().numpy()
            p1, p2 = got[0, : len(mine)], got[1, : len(mine)]
        full1[mine], full2[mine] = p1, p2
    # ---- output (rank 0), src/call.rs:137-157: BED order for -t 1, (human chrom, start) order otherwise.  One call into the
    # host library on the f64 arrays (the code inq_genotype_repeats itself ends with): 500 000 rows take tens of milliseconds
    out = sys.stdout if out is None else out
    t0 = time.perf_counter()
    run.write_inq(full1, full2, out)
    if stats is not None:
        stats["output_s"] = time.perf_counter() - t0
        stats["rows"] = n
    run.close()


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="inquistr_amd.call_dist", description="inquiSTR call, one process per GPU")
    ap.add_argument("bam")
    ap.add_argument("-r", "--region")
    ap.add_argument("-R", "--region-file", "--region_file", dest="region_file")
    ap.add_argument("-m", "--minlen", type=int, default=5)
    ap.add_argument("-s", "--support", type=int, default=3)
    ap.add_argument("-t", "--threads", type=int, default=1)
    ap.add_argument("-u", "--unphased", action="store_true")
    ap.add_argument("--sample-name", "--sample_name", dest="sample_name")
    ap.add_argument("-o", "--output", help="write the .inq here instead of stdout (gloo prints connection notes on stdout)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--same-device", action="store_true", help="all ranks on device 0 (rehearsal on a one-GPU box)")
    ap.add_argument("--frontend", default=None, choices=["host", "device"], help="default: by the amount of BAM each rank reads")
    ap.add_argument("--rows", default=None, choices=["device", "host"], help="where a rank's rows wait for the gather (default: device memory on nccl)")
    ap.add_argument("--stats-dir", default=None, help="every rank leaves rank<r>.json there: its loci, seconds, BAM bytes read, the gather's time")
    a = ap.parse_args(argv)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    device = 0 if a.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            torch.cuda.set_device(device)
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")
    out = open(a.output, "w") if (a.output and rank == 0) else None
    stats = {} if a.stats_dir else None
    try:
        genotype_repeats_distributed(a.bam, a.region, a.region_file, a.minlen, a.support, a.threads, a.unphased,
                                     a.sample_name, out=out, rank=rank, world=world, device=device, frontend=a.frontend, stats=stats, rows=a.rows)
        if stats is not None:
            import json

            os.makedirs(a.stats_dir, exist_ok=True)
            with open(os.path.join(a.stats_dir, f"rank{rank}.json"), "w") as f:
                json.dump(stats, f)
    except hostcall.CallError as e:  # the same status on every rank (the failure was exchanged before the gather)
        if rank == 0:
            print(e.message, file=sys.stderr)
        return e.status
    finally:
        if out is not None:
            out.close()
        if world > 1 and dist.is_initialized():
            dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
