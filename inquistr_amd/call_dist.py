"""`inquiSTR call` over several GPUs of one node: one process per GPU (torch.distributed; backend
`nccl` = RCCL on the GPUs, `gloo` for CPU rehearsal), loci partitioned by rank, no data-path
collective; the only exchange is the gather of the per-shard rows to rank 0, which writes the `.inq`.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        -m inquistr_amd.call_dist sample.bam -R loci.bed -t 8 > sample.inq

Each rank runs the C++ driver on its own contiguous slice of the (contig, start)-sorted targets: it reads
only the part of the BAM that slice needs and inflates, scans and calls it on its own GPU (device front
end), or sweeps it on the host for small inputs; the reference's counterpart is the
rayon loop over loci (src/call.rs:115-136), which shares nothing but the output Vec.
"""
from __future__ import annotations

import argparse
import functools
import os
import sys
import tempfile
from typing import Callable, List, Optional

import numpy as np

from . import call as hostcall


def _cuts_by_file_bytes(bamp: str, sorted_targets, world: int) -> List[int]:
    """world + 1 cut points into the sorted target list.  Cost of a target = compressed bytes between its
    scan start and the next target's (same contig), capped so that one far-away locus does not own a whole
    chromosome; falls back to equal counts when the index gives nothing."""
    n = len(sorted_targets)
    L = hostcall.load()
    bai = bamp + ".bai" if os.path.exists(bamp + ".bai") else os.path.splitext(bamp)[0] + ".bai"
    tid_of = {}
    offs = np.zeros(n, dtype=np.float64)
    for k, (chrom, start, _end) in enumerate(sorted_targets):
        if chrom not in tid_of:
            tid_of[chrom] = L.inq_host_bam_tid(os.fspath(bamp).encode(), chrom.encode())
        offs[k] = L.inq_host_bai_file_offset(bai.encode(), tid_of[chrom], max(0, start - 10))
    cost = np.ones(n, dtype=np.float64)
    if n > 1 and offs.max() > 0:
        d = np.diff(offs)
        ok = d > 0
        if ok.any():
            cap = np.percentile(d[ok], 99) * 4 + 1
            cost[:-1] += np.clip(np.where(ok, d, 0), 0, cap)
            cost[-1] += np.median(d[ok])
    csum = np.concatenate([[0.0], np.cumsum(cost)])
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(min(max(np.searchsorted(csum, csum[-1] * r / world, side="left"), cuts[-1]), n)))
    cuts.append(n)
    return cuts


def genotype_repeats_distributed(bamp: str, region: Optional[str], region_file: Optional[str], minlen: int = 5,
                                 support: int = 3, threads: int = 1, unphased: bool = False,
                                 sample_name: Optional[str] = None, out=None, rank: int = 0, world: int = 1,
                                 device: int = 0, compute: Optional[Callable] = None, group=None,
                                 frontend: Optional[str] = None) -> None:
    """Same arguments as call.genotype_repeats plus (rank, world).  Rank 0 writes header + rows."""
    import torch
    import torch.distributed as dist

    # every rank parses and validates the full target list exactly like a single-process run would
    fe_all = hostcall.FrontEnd(bamp, region=region, region_file=region_file, minlen=minlen, support=support,
                               threads=1, unphased=unphased, sample_name=sample_name)
    targets = fe_all.targets()
    sample = fe_all.sample
    fe_all.close()
    n = len(targets)
    # contiguous slices of the position-sorted list, cut so that every rank has about the same amount of
    # BAM to read (the .bai linear index gives the file offset of every target; SURVEY.md §8e asks for shards
    # balanced by work, not by locus count).  Every rank computes the same cuts.
    order = sorted(range(n), key=lambda i: (targets[i][0], targets[i][1], i))
    cuts = _cuts_by_file_bytes(bamp, [targets[i] for i in order], world)
    lo, hi = cuts[rank], cuts[rank + 1]
    mine = order[lo:hi]
    p1 = np.full(len(mine), np.nan)
    p2 = np.full(len(mine), np.nan)
    if mine:
        with tempfile.NamedTemporaryFile("w", suffix=".bed", delete=False) as f:
            for i in mine:
                f.write(f"{targets[i][0]}\t{targets[i][1]}\t{targets[i][2]}\n")
            sub_bed = f.name
        try:
            if compute is None:
                # the product path: the C++ driver on this rank's slice, which picks the device front end
                # (inflate + record scan + join on this rank's GPU) or the host sweep by the amount of BAM.
                # The rows come back as text; integers, halves and NaN survive that exactly.
                with tempfile.NamedTemporaryFile("w+", suffix=".inq", delete=False) as rows:
                    rows_path = rows.name
                try:
                    with open(rows_path, "w") as rf:
                        hostcall.genotype_repeats(bamp, None, sub_bed, minlen, support, threads, unphased, sample_name, None,
                                                  out=rf, device=device, frontend=frontend)
                    got = {}
                    with open(rows_path) as rf:
                        next(rf)
                        for line in rf:
                            c, s0, e0, a, b = line.rstrip("\n").split("\t")
                            got[(c, int(s0), int(e0))] = (float(a), float(b))
                finally:
                    os.unlink(rows_path)
                for k, i in enumerate(mine):
                    p1[k], p2[k] = got[tuple(targets[i])]
            else:  # tests: per-batch compute supplied by the caller (the oracle, on CPU-only machines)
                fe = hostcall.FrontEnd(bamp, region_file=sub_bed, minlen=minlen, support=support, threads=threads,
                                       unphased=unphased, sample_name=sample_name)
                for batch, idx in fe.batches():
                    a, b = compute(batch)
                    p1[idx], p2[idx] = a, b
                fe.close()
        finally:
            os.unlink(sub_bed)
    if world > 1:
        width = max(cuts[r + 1] - cuts[r] for r in range(world))
        gdev = torch.device("cuda", device) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        buf = torch.full((2, max(width, 1)), float("nan"), dtype=torch.float64, device=gdev)
        buf[0, : len(mine)] = torch.from_numpy(p1)
        buf[1, : len(mine)] = torch.from_numpy(p2)
        bufs = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
        dist.gather(buf, bufs, dst=0, group=group)
        if rank != 0:
            return
        full1, full2 = np.full(n, np.nan), np.full(n, np.nan)
        for r in range(world):
            sl = order[cuts[r] : cuts[r + 1]]
            full1[sl] = bufs[r][0, : len(sl)].cpu().numpy()
            full2[sl] = bufs[r][1, : len(sl)].cpu().numpy()
    else:
        full1, full2 = np.full(n, np.nan), np.full(n, np.nan)
        full1[mine], full2[mine] = p1, p2
    # output, src/call.rs:137-157: BED order for -t 1, (human chrom, start) order otherwise
    L = hostcall.load()
    import ctypes as C

    rows = list(range(n))
    if threads > 1:
        def cmp(x, y):
            c = L.inq_host_human_compare(targets[x][0].encode(), targets[y][0].encode())
            return c or (targets[x][1] > targets[y][1]) - (targets[x][1] < targets[y][1])

        rows.sort(key=functools.cmp_to_key(cmp))
    out = sys.stdout if out is None else out
    buf = C.create_string_buffer(4096)
    L.inq_host_format_header(sample.encode(), buf, len(buf))
    lines = [buf.value.decode()]
    for i in rows:
        L.inq_host_format_row(targets[i][0].encode(), targets[i][1], targets[i][2], float(full1[i]), float(full2[i]), buf, len(buf))
        lines.append(buf.value.decode())
    out.write("\n".join(lines) + "\n")
    out.flush()


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
    try:
        genotype_repeats_distributed(a.bam, a.region, a.region_file, a.minlen, a.support, a.threads, a.unphased,
                                     a.sample_name, out=out, rank=rank, world=world, device=device, frontend=a.frontend)
    except hostcall.CallError as e:
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
