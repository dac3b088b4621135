"""Seeded synthetic pile-ups for the workloads BASELINE.json names.

A counter-based integer hash drives everything, written once over an array namespace so that
numpy (host, feeds the CPU oracle) and torch (device, feeds bench.py) produce bit-identical
buffers for the same (workload, seed, locus range).  The generator is measurement/test
plumbing; it never computes a call.

Workloads
  phased10k   #2  10k loci x 30 reads x ~200 ops, HP 1/2 alternating, phased
  unphased100k #3 100k loci x 30 reads x ~200 ops, --unphased (the HBM roofline run)
  shard500k   #4  500k loci, same distribution as #3 (sharded by the caller)
  expansion50k #5 50k loci, 10 % of reads with one 5-50 kb insertion inside the window and
                  ~2000 ops, 5 % of reads soft-clipped inside the window, phased
Read layout: op index even = M (len 5..200), odd = I or D with a geometric(p=0.3) length (about
one in six exceeds the default minlen of 5); 60 % of (locus, haplotype) alleles differ from the
reference by a 6..300 bp insertion or 6..60 bp deletion that every read of the haplotype carries
inside the window, so the medians are not trivially 0.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np

from .batch import READ_DTYPE, Batch

_GEOM_P = 0.3
# quantile table of a geometric(0.3) length, indexed by one hash byte (exact, no float at run time)
_GEOM = np.minimum(1 + np.floor(np.log1p(-(np.arange(256) + 0.5) / 256.0) / np.log(1.0 - _GEOM_P)), 60).astype(np.int64)


@dataclass(frozen=True)
class Workload:
    name: str
    n_loci: int
    reads_per_locus: int = 30
    unphased: bool = False
    heavy_pct: int = 0
    clip_pct: int = 0
    seed: int = 1
    minlen: int = 5
    support: int = 3


WORKLOADS: Dict[str, Workload] = {
    "phased10k": Workload("phased10k", 10_000, seed=1),
    "unphased100k": Workload("unphased100k", 100_000, unphased=True, seed=2),
    "shard500k": Workload("shard500k", 500_000, unphased=True, seed=3),
    "expansion50k": Workload("expansion50k", 50_000, heavy_pct=10, clip_pct=5, seed=4),
    # not a BASELINE config: every read ONT-like (~2000-2400 ops, 8-10 chunks per read) — exercises the
    # chunk pipeline inside long reads rather than across short ones
    "longreads20k": Workload("longreads20k", 20_000, heavy_pct=100, seed=5),
}


class _NP:
    """numpy namespace adapter"""

    name = "numpy"

    def __init__(self):
        self.geom = _GEOM

    def arange(self, a, b=None):
        return np.arange(a, b, dtype=np.int64) if b is not None else np.arange(a, dtype=np.int64)

    where = staticmethod(np.where)

    def cumsum0(self, x):  # exclusive cumsum, 1-D
        out = np.zeros(x.shape[0] + 1, dtype=np.int64)
        np.cumsum(x, out=out[1:])
        return out

    def take(self, table, idx):
        return table[idx]

    def sum1(self, x):
        return x.sum(axis=1)

    def maximum0(self, x):
        return np.maximum(x, 0)

    def select(self, x, mask):
        return x[mask]

    def repeat(self, x, n):
        return np.repeat(x, n)

    def stack4(self, a, b, c, d):
        return np.stack([a, b, c, d], axis=1)

    def zeros(self, n):
        return np.zeros(n, dtype=np.int64)


class _Torch:
    """torch namespace adapter (device tensors, int64 arithmetic)"""

    name = "torch"

    def __init__(self, device):
        import torch

        self.t = torch
        self.device = device
        self.geom = torch.from_numpy(_GEOM).to(device)

    def arange(self, a, b=None):
        t = self.t
        return t.arange(a, b, dtype=t.int64, device=self.device) if b is not None else t.arange(a, dtype=t.int64, device=self.device)

    def where(self, c, a, b):
        t = self.t
        if not t.is_tensor(a):
            a = t.tensor(a, dtype=t.int64, device=self.device)
        if not t.is_tensor(b):
            b = t.tensor(b, dtype=t.int64, device=self.device)
        return t.where(c, a, b)

    def cumsum0(self, x):
        t = self.t
        out = t.zeros(x.shape[0] + 1, dtype=t.int64, device=self.device)
        t.cumsum(x, 0, out=out[1:])
        return out

    def take(self, table, idx):
        return table[idx]

    def sum1(self, x):
        return x.sum(dim=1)

    def maximum0(self, x):
        return self.t.clamp_min(x, 0)

    def select(self, x, mask):
        return x[mask]

    def repeat(self, x, n):
        return self.t.repeat_interleave(x, n)

    def stack4(self, a, b, c, d):
        return self.t.stack([a, b, c, d], dim=1)

    def zeros(self, n):
        return self.t.zeros(n, dtype=self.t.int64, device=self.device)


def _h(x, salt: int, seed: int):
    """lowbias32 of a 64-bit counter folded to 32 bits; int64 containers, identical in numpy/torch."""
    x = x + (salt * 0x9E3779B1 + seed * 0x85EBCA77)
    x = (x ^ (x >> 32)) & 0xFFFFFFFF
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x = x ^ (x >> 16)
    return x


def _gen_chunk(xp, wl: Workload, g0: int, g1: int, off4_base: int):
    """Loci [g0, g1) of the workload.  Returns dict of int64 arrays/tensors."""
    R, seed = wl.reads_per_locus, wl.seed
    L = g1 - g0
    g = xp.arange(g0, g1)
    li = g % 10000
    start = 50_000 + 20_000 * li + _h(g, 1, seed) % 64
    end = start + 20 + _h(g, 2, seed) % 181
    k = xp.arange(R)
    r = (g[:, None] * R + k[None, :]).reshape(-1)
    jloc = xp.repeat(xp.arange(L), R)
    kk = r % R
    heavy = (_h(r, 4, seed) % 100) < wl.heavy_pct
    clipr = (~heavy) & ((_h(r, 7, seed) % 100) < wl.clip_pct)
    n_ops = xp.where(heavy, 2000 + _h(r, 3, seed) % 401, 180 + _h(r, 3, seed) % 41)
    maxops = 2400 if wl.heavy_pct else 220
    t = xp.arange(maxops)[None, :]
    x = r[:, None] * 4096 + t
    odd = (t & 1) == 1
    is_del = (_h(x, 8, seed) & 1) == 1
    ilen = xp.take(xp.geom, _h(x, 9, seed) % 256)
    op = xp.where(odd & is_del, 2, xp.where(odd, 1, 0))
    ln = xp.where(odd, ilen, 5 + _h(x, 6, seed) % 196)
    # every non-clipped read carries its haplotype's allele as ONE op placed inside the window:
    # op index t_a (odd, 41..79) becomes an I/D of the allele length (+0..2 bp of read noise)
    t_a = 2 * (_h(r, 10, seed) % 20) + 41
    ga = g[jloc] * 2 + (kk & 1)
    has_allele = (_h(ga, 15, seed) % 100) < 60
    a_del = (_h(ga, 17, seed) & 1) == 1
    a_len = xp.where(a_del, 6 + _h(ga, 16, seed) % 55, 6 + _h(ga, 16, seed) % 295) + _h(r, 18, seed) % 3
    at = t == t_a[:, None]
    hit = at & has_allele[:, None]
    op = xp.where(hit, xp.where(a_del, 2, 1)[:, None], op)
    ln = xp.where(hit, a_len[:, None], ln)
    if wl.heavy_pct:
        big = heavy[:, None] & at
        op = xp.where(big, 1, op)
        ln = xp.where(big, (5000 + _h(r, 11, seed) % 45001)[:, None], ln)
    if wl.clip_pct:
        c0 = clipr[:, None] & (t == 0)
        op = xp.where(c0, 4, op)
        ln = xp.where(c0, (20 + _h(r, 12, seed) % 1981)[:, None], ln)
    valid = t < n_ops[:, None]
    word = xp.where(valid, ln * 16 + op, 0)
    se = (start - 10)[jloc]
    width = (end + 10)[jloc] - se
    consumed = xp.where(valid & ((op == 0) | (op == 2)), ln, 0)
    prefix = xp.sum1(xp.where(t < t_a[:, None], consumed, 0))
    # reference_position at op t_a = pos + 1 + prefix = se + 1 + u, u in [0, width-2]: inside the window
    pos = se + _h(r, 13, seed) % (width - 1) - prefix
    if wl.clip_pct:
        pos = xp.where(clipr, se + _h(r, 14, seed) % (width - 2), pos)
    pos = xp.maximum0(pos)
    n4 = (n_ops + 3) // 4
    off4 = xp.cumsum0(n4)
    total4 = int(off4[-1])
    padded = t < (4 * n4)[:, None]
    cigar = xp.select(word, padded)
    bits = 4  # INQ_READ_HAS_HP
    misc = 60 + (bits << 8) + ((1 + (kk & 1)) << 16)
    reads4 = xp.stack4(off4[:-1] + off4_base, n_ops, pos, misc)
    return {
        "cigar": cigar,
        "reads4": reads4,
        "locus_start": start,
        "locus_end": end,
        "total4": total4,
        "n_reads": L * R,
    }


def _chunk_loci(wl: Workload) -> int:
    return 256 if wl.heavy_pct else 4096


def generate_numpy(wl: Workload, lo: int = 0, hi: int = None) -> Batch:
    """Loci [lo, hi) of the workload as a host Batch (for the oracle and host-entry tests)."""
    hi = wl.n_loci if hi is None else hi
    xp = _NP()
    cig, rds, ls, le = [], [], [], []
    off4 = 0
    step = _chunk_loci(wl)
    for g0 in range(lo, hi, step):
        c = _gen_chunk(xp, wl, g0, min(hi, g0 + step), off4)
        off4 += c["total4"]
        cig.append(c["cigar"].astype(np.uint32))
        rds.append(c["reads4"])
        ls.append(c["locus_start"])
        le.append(c["locus_end"])
    n_loci = hi - lo
    R = wl.reads_per_locus
    r4 = np.concatenate(rds) if rds else np.zeros((0, 4), dtype=np.int64)
    reads = np.zeros(r4.shape[0], dtype=READ_DTYPE)
    reads["cigar_off4"] = r4[:, 0]
    reads["n_cigar"] = r4[:, 1]
    reads["pos"] = r4[:, 2]
    reads["mapq"] = r4[:, 3] & 0xFF
    reads["bits"] = (r4[:, 3] >> 8) & 0xFF
    reads["phase"] = (r4[:, 3] >> 16) & 0xFF
    return Batch(
        cigar=np.concatenate(cig) if cig else np.zeros(0, dtype=np.uint32),
        reads=reads,
        pair_read=np.arange(n_loci * R, dtype=np.uint32),
        locus_pair_off=(np.arange(n_loci + 1, dtype=np.uint64) * np.uint64(R)),
        locus_start=(np.concatenate(ls) if ls else np.zeros(0)).astype(np.uint32),
        locus_end=(np.concatenate(le) if le else np.zeros(0)).astype(np.uint32),
        minlen=wl.minlen,
        support=wl.support,
        unphased=wl.unphased,
    )


class DeviceBatch:
    """The same buffers resident in HBM as torch tensors + the inq_batch_t / inq_result_t that
    point at them (device pointers)."""

    def __init__(self, wl: Workload, device, lo: int = 0, hi: int = None, debug: bool = False, neighbors: int = 0):
        """neighbors = k also offers every locus the reads of its k neighbours on each side (in file
        order).  Those reads do not overlap the locus, so htslib's overlap rule on the device drops
        them: results must not change, while every CIGAR is now walked 2k+1 times (shared reads)."""
        import torch

        from .batch import InqBatchC, InqResultC

        hi = wl.n_loci if hi is None else hi
        xp = _Torch(device)
        cig, rds, ls, le = [], [], [], []
        off4 = 0
        step = _chunk_loci(wl)
        for g0 in range(lo, hi, step):
            c = _gen_chunk(xp, wl, g0, min(hi, g0 + step), off4)
            off4 += c["total4"]
            cig.append(c["cigar"].to(torch.int32))
            rds.append(c["reads4"].to(torch.int32))
            ls.append(c["locus_start"].to(torch.int32))
            le.append(c["locus_end"].to(torch.int32))
        self.wl, self.lo, self.hi = wl, lo, hi
        self.n_loci = hi - lo
        R = wl.reads_per_locus
        self.n_reads = self.n_loci * R
        self.cigar = torch.cat(cig).contiguous()
        self.reads = torch.cat(rds).contiguous()  # [n_reads, 4] int32 == inq_read_t
        if neighbors == 0:
            self.n_pairs = self.n_reads
            self.pair_read = torch.arange(self.n_pairs, dtype=torch.int32, device=device)
            self.locus_pair_off = torch.arange(self.n_loci + 1, dtype=torch.int64, device=device) * R
        else:
            j = torch.arange(self.n_loci, dtype=torch.int64, device=device)
            first = torch.clamp(j - neighbors, min=0) * R
            last = torch.clamp(j + neighbors + 1, max=self.n_loci) * R
            cnt = last - first
            self.locus_pair_off = torch.zeros(self.n_loci + 1, dtype=torch.int64, device=device)
            torch.cumsum(cnt, 0, out=self.locus_pair_off[1:])
            self.n_pairs = int(self.locus_pair_off[-1].item())
            within = torch.arange(self.n_pairs, dtype=torch.int64, device=device) - torch.repeat_interleave(self.locus_pair_off[:-1], cnt)
            self.pair_read = (torch.repeat_interleave(first, cnt) + within).to(torch.int32)
        self.locus_start = torch.cat(ls).contiguous()
        self.locus_end = torch.cat(le).contiguous()
        self.phase1 = torch.full((self.n_loci,), float("nan"), dtype=torch.float64, device=device)
        self.phase2 = torch.full((self.n_loci,), float("nan"), dtype=torch.float64, device=device)
        self.pair_call = torch.zeros(self.n_pairs, dtype=torch.int64, device=device) if debug else None
        self.pair_bits = torch.zeros(self.n_pairs, dtype=torch.uint8, device=device) if debug else None
        self.n_ops_total = int(self.reads[:, 1].to(torch.int64)[self.pair_read.to(torch.int64)].sum().item())  # over pairs

        b = InqBatchC()
        b.n_reads, b.n_cigar_words = self.n_reads, int(self.cigar.numel())
        b.n_pairs, b.n_loci = self.n_pairs, self.n_loci
        b.cigar, b.reads = self.cigar.data_ptr(), self.reads.data_ptr()
        b.pair_read, b.locus_pair_off = self.pair_read.data_ptr(), self.locus_pair_off.data_ptr()
        b.locus_start, b.locus_end = self.locus_start.data_ptr(), self.locus_end.data_ptr()
        b.minlen, b.support, b.unphased, b.reserved = wl.minlen, wl.support, int(wl.unphased), 0
        self.c_batch = b
        r = InqResultC()
        r.phase1, r.phase2 = self.phase1.data_ptr(), self.phase2.data_ptr()
        r.pair_call = self.pair_call.data_ptr() if debug else None
        r.pair_bits = self.pair_bits.data_ptr() if debug else None
        self.c_result = r

    def algorithmic_bytes(self) -> int:
        return 4 * self.n_ops_total + 20 * self.n_pairs + 32 * self.n_loci

    def ops_per_locus(self):
        """CIGAR ops each locus makes the device walk (sum over its pairs): the cost inquistr_amd.shard balances by."""
        import torch

        ops = self.reads[:, 1].to(torch.int64)[self.pair_read.to(torch.int64)]
        csum = torch.zeros(self.n_pairs + 1, dtype=torch.int64, device=ops.device)
        torch.cumsum(ops, 0, out=csum[1:])
        return csum[self.locus_pair_off[1:]] - csum[self.locus_pair_off[:-1]]
