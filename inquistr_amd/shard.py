"""Locus sharding for one-process-per-GPU runs, and the gather of per-shard result rows.

Loci are independent in the reference (`repeats.par_bridge().for_each`, src/call.rs:115-118: the
only shared state is the output Vec), so the path shards with no data-path collective.  Rank r
takes a contiguous locus range balanced by CIGAR-op count (a deep or long-read region would
otherwise leave one GPU working while seven idle) plus the reads those loci reference; the one
exchange step is the gather of 2 x f64 per locus to rank 0 (torch.distributed: `nccl` = RCCL on
the GPUs, `gloo` in the CPU tests).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from .batch import Batch


def balanced_ranges(cost_per_locus: np.ndarray, world: int) -> List[Tuple[int, int]]:
    """Contiguous [lo, hi) ranges, one per rank, with near-equal total cost.  Every locus lands in
    exactly one range; ranges may be empty when there are fewer loci than ranks."""
    n = int(cost_per_locus.shape[0])
    if world <= 0:
        raise ValueError("world must be positive")
    csum = np.concatenate([[0], np.cumsum(cost_per_locus.astype(np.float64) + 1.0)])
    total = csum[-1]
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        k = int(np.searchsorted(csum, target, side="left"))
        cuts.append(min(max(k, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def locus_cost(batch: Batch) -> np.ndarray:
    """CIGAR ops each locus makes the device walk (Σ over its pairs)."""
    ops = batch.cigar_ops_per_pair()
    csum = np.concatenate([[0], np.cumsum(ops)])
    off = batch.locus_pair_off.astype(np.int64)
    return csum[off[1:]] - csum[off[:-1]]


def shard_batch(batch: Batch, rank: int, world: int) -> Tuple[Batch, List[Tuple[int, int]]]:
    ranges = balanced_ranges(locus_cost(batch), world)
    lo, hi = ranges[rank]
    return batch.slice_loci(lo, hi), ranges


def pack_rows(p1, p2, width: int):
    """This rank's rows as the rectangular [2, width] f64 buffer the gather moves (NaN-padded)."""
    import torch

    mine = torch.full((2, max(width, 1)), float("nan"), dtype=torch.float64, device=p1.device)
    n = p1.shape[0]
    mine[0, :n] = p1
    mine[1, :n] = p2
    return mine


def unpack_rows(bufs, ranges: Sequence[Tuple[int, int]]):
    """Rank 0: the gathered buffers (one per rank, in rank order) back into locus order."""
    out1 = np.concatenate([bufs[r][0, : ranges[r][1] - ranges[r][0]].cpu().numpy() for r in range(len(ranges))])
    out2 = np.concatenate([bufs[r][1, : ranges[r][1] - ranges[r][0]].cpu().numpy() for r in range(len(ranges))])
    return out1, out2


def gather_rows(p1, p2, ranges: Sequence[Tuple[int, int]], rank: int, world: int, group=None):
    """Gathers the shards' (phase1, phase2) rows to rank 0 in locus order.  p1/p2 are this rank's
    rows as torch tensors (CPU for gloo, device for nccl).  Returns (phase1, phase2) numpy arrays on
    rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return p1.cpu().numpy(), p2.cpu().numpy()
    mine = pack_rows(p1, p2, max(hi - lo for lo, hi in ranges))
    bufs = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, bufs, dst=0, group=group)
    if rank != 0:
        return None
    return unpack_rows(bufs, ranges)


def run_sharded(batch: Batch, compute: Callable[[Batch], Tuple[np.ndarray, np.ndarray]], rank: int, world: int,
                device: Optional[str] = None, group=None):
    """Shard -> compute on this rank's shard -> gather.  `compute` is the HIP path in production
    (inquistr_amd.call); tests inject a CPU stand-in to exercise the plumbing on gloo."""
    import torch

    sub, ranges = shard_batch(batch, rank, world)
    a, b = compute(sub) if sub.n_loci else (np.zeros(0), np.zeros(0))
    dev = torch.device(device) if device else torch.device("cpu")
    return gather_rows(torch.from_numpy(np.ascontiguousarray(a)).to(dev), torch.from_numpy(np.ascontiguousarray(b)).to(dev),
                       ranges, rank, world, group)
