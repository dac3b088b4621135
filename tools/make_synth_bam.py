#!/usr/bin/env python3
"""Writes a synthetic workload (inquistr_amd/synth.py) as a coordinate-sorted BAM + .bai + BED:
the end-to-end (L2) form of BASELINE.json's configs.  Measurement plumbing, never the product.
10 000 loci per contig (chr1, chr2, ...), reads carry HP:C, SEQ is '*' (l_seq = 0).
usage: tools/make_synth_bam.py <workload> <n_loci> <out_prefix> [seq]   (seq: records with SEQ / QUAL / MM / ML like a real long-read BAM)"""
from __future__ import annotations

import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inquistr_amd import synth  # noqa: E402
from tools import bamio  # noqa: E402

LOCI_PER_CONTIG = 10_000
CONTIG_LEN = 50_000 + 20_000 * LOCI_PER_CONTIG + 400_000

FIXED = np.dtype([("block_size", "<i4"), ("refID", "<i4"), ("pos", "<i4"), ("l_read_name", "u1"), ("mapq", "u1"),
                  ("bin", "<u2"), ("n_cigar", "<u2"), ("flag", "<u2"), ("l_seq", "<i4"), ("next_refID", "<i4"),
                  ("next_pos", "<i4"), ("tlen", "<i4"), ("name", "S12")])
assert FIXED.itemsize == 48


def reg2bin_vec(beg, end):
    end = end - 1
    out = np.zeros_like(beg)
    done = np.zeros(beg.shape, dtype=bool)
    for shift, first in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        m = ~done & ((beg >> shift) == (end >> shift))
        out[m] = first + (beg[m] >> shift)
        done |= m
    return out


def records_for(batch, tid: int, first_read_id: int):
    """(bytes of all records of the batch sorted by pos, per-record (size, beg, end))"""
    r = batch.reads
    n = len(r)
    ncig = r["n_cigar"].astype(np.int64)
    off = r["cigar_off4"].astype(np.int64) * 4
    w = batch.cigar.astype(np.int64)
    consumed = np.where(np.isin(w & 15, (0, 2, 3, 7, 8)), w >> 4, 0)
    span = np.add.reduceat(consumed, off) if n else np.zeros(0, dtype=np.int64)
    pos = r["pos"].astype(np.int64)
    end = pos + np.maximum(span, 1)
    order = np.argsort(pos, kind="stable")
    sizes = 48 + 4 * ncig + 4
    so = sizes[order]
    starts = np.concatenate([[0], np.cumsum(so)])[:-1]
    out = np.zeros(int(so.sum()), dtype=np.uint8)
    fx = np.zeros(n, dtype=FIXED)
    fx["block_size"] = so - 4
    fx["refID"] = tid
    fx["pos"] = pos[order]
    fx["l_read_name"] = 12
    fx["mapq"] = r["mapq"][order]
    fx["bin"] = reg2bin_vec(pos[order], end[order])
    fx["n_cigar"] = ncig[order]
    fx["flag"] = 0
    fx["l_seq"] = 0
    fx["next_refID"] = -1
    fx["next_pos"] = -1
    ids = first_read_id + order
    digits = ((ids[:, None] // 10 ** np.arange(9, -1, -1)) % 10 + 48).astype(np.uint8)
    names = np.concatenate([np.full((n, 1), ord("r"), np.uint8), digits, np.zeros((n, 1), np.uint8)], axis=1)
    fx["name"] = names.view("S12").reshape(n)
    out[(starts[:, None] + np.arange(48)).reshape(-1)] = fx.view(np.uint8).reshape(-1)
    # CIGAR words
    nc_o = ncig[order]
    tot = int(nc_o.sum())
    within = np.arange(tot) - np.repeat(np.concatenate([[0], np.cumsum(nc_o)])[:-1], nc_o)
    src = np.repeat(off[order], nc_o) + within
    dst = np.repeat(starts + 48, nc_o) + 4 * within
    words = batch.cigar[src].astype("<u4").view(np.uint8).reshape(-1, 4)
    out[(dst[:, None] + np.arange(4)).reshape(-1)] = words.reshape(-1)
    # HP:C:<phase>
    aux = np.zeros((n, 4), dtype=np.uint8)
    aux[:, 0], aux[:, 1], aux[:, 2] = ord("H"), ord("P"), ord("C")
    aux[:, 3] = r["phase"][order]
    out[((starts + 48 + 4 * nc_o)[:, None] + np.arange(4)).reshape(-1)] = aux.reshape(-1)
    return out.tobytes(), so, pos[order], end[order]


def records_with_seq(batch, tid: int, first_read_id: int, rng: np.random.Generator):
    """Like records_for, but every record carries what a real long-read BAM does around the CIGAR: SEQ (random
    bases) and QUAL (random Phred 0..50) of the read's query length, a methylation-style ML:B,C array and MM:Z
    string, and HP:C as the LAST tag (where phasing tools append it).  Record by record: ~30 KB each."""
    r = batch.reads
    n = len(r)
    ncig = r["n_cigar"].astype(np.int64)
    off = r["cigar_off4"].astype(np.int64) * 4
    w = batch.cigar.astype(np.int64)
    consumed = np.where(np.isin(w & 15, (0, 2, 3, 7, 8)), w >> 4, 0)
    query = np.where(np.isin(w & 15, (0, 1, 4, 7, 8)), w >> 4, 0)
    span = np.add.reduceat(consumed, off) if n else np.zeros(0, dtype=np.int64)
    qlen = np.add.reduceat(query, off) if n else np.zeros(0, dtype=np.int64)
    pos = r["pos"].astype(np.int64)
    end = pos + np.maximum(span, 1)
    order = np.argsort(pos, kind="stable")
    bins = reg2bin_vec(pos, end)
    parts, sizes = [], []
    for k in order:
        l_seq = int(min(qlen[k], 60_000))
        name = b"r%010d\0" % (first_read_id + int(k))
        words = batch.cigar[off[k] : off[k] + ncig[k]].astype("<u4").tobytes()
        seq = rng.integers(0, 256, (l_seq + 1) // 2, dtype=np.uint8).tobytes()
        qual = rng.integers(0, 51, l_seq, dtype=np.uint8).tobytes()
        n_mod = l_seq // 25
        ml = b"MLBC" + struct.pack("<I", n_mod) + rng.integers(0, 256, n_mod, dtype=np.uint8).tobytes()
        mm = b"MMZC+m," + b",".join(b"%d" % v for v in rng.integers(0, 40, n_mod)) + b";\0"
        aux = b"NMi" + struct.pack("<i", 17) + ml + mm + b"HPC" + bytes([int(r["phase"][k])])
        core = struct.pack("<iiBBHHHIiii", tid, int(pos[k]), len(name), int(r["mapq"][k]), int(bins[k]), int(ncig[k]), 0, l_seq, -1, -1, 0)
        body = core + name + words + seq + qual + aux
        parts.append(struct.pack("<I", len(body)) + body)
        sizes.append(len(body) + 4)
    return b"".join(parts), np.array(sizes, dtype=np.int64), pos[order], end[order]


def write(workload: str, n_loci: int, prefix: str, level: int = 1, seq: bool = False):
    wl = synth.WORKLOADS[workload]
    n_contigs = (n_loci + LOCI_PER_CONTIG - 1) // LOCI_PER_CONTIG
    refs = [(f"chr{c + 1}", CONTIG_LEN) for c in range(n_contigs)]
    w = bamio.BamWriter(prefix + ".bam", refs, level=level)
    bed = open(prefix + ".bed", "w")
    step = 256 if wl.heavy_pct else 2000
    if seq:
        step = 64
    rng = np.random.default_rng(12345)
    rid = 0
    for c in range(n_contigs):
        c_lo, c_hi = c * LOCI_PER_CONTIG, min(n_loci, (c + 1) * LOCI_PER_CONTIG)
        for g0 in range(c_lo, c_hi, step):
            b = synth.generate_numpy(wl, g0, min(c_hi, g0 + step))
            blob, sizes, beg, end = records_with_seq(b, c, rid, rng) if seq else records_for(b, c, rid)
            u0 = len(w.buf)
            w.buf += blob
            offs = u0 + np.concatenate([[0], np.cumsum(sizes)])
            w.recs.extend(zip(offs[:-1].tolist(), offs[1:].tolist(), [c] * len(sizes), beg.tolist(), end.tolist(), [0] * len(sizes)))
            rid += len(sizes)
            for s, e in zip(b.locus_start.tolist(), b.locus_end.tolist()):
                bed.write(f"chr{c + 1}\t{s}\t{e}\n")
    bed.close()
    w.close()
    return rid


_NATIVE = None


def native_lib():
    """tools/libsynthbam.so (tools/synth_bam_writer.cc), built on first use."""
    global _NATIVE
    if _NATIVE is None:
        import ctypes as C
        import subprocess

        here = os.path.dirname(os.path.abspath(__file__))
        so, src = os.path.join(here, "libsynthbam.so"), os.path.join(here, "synth_bam_writer.cc")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", so, src, "-lz"])
        L = C.CDLL(so)
        L.inq_synth_write_bam.restype = C.c_int
        L.inq_synth_write_bam.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_uint32, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
        L.inq_synth_write_bam_seq.restype = C.c_int
        L.inq_synth_write_bam_seq.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64),
                                              C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]
        _NATIVE = L
    return _NATIVE


def write_native(workload: str, n_loci: int, prefix: str, level: int = 1, threads: int = 0, device=None, seq: bool = False,
                 qual_mode: int = 1, seed: int = 12345, slab_blocks: int = 0, info: dict = None):
    """Same files as write() (byte for byte, tests/test_synth_bam_native.py), records assembled, deflated and indexed by
    native threads.  device: a torch device to generate the workload on (bit-identical to the numpy generator).
    seq: records shaped like a real long-read BAM (SEQ + QUAL of the query length, NM, ML / MM, HP last; see
    inq_synth_write_bam_seq: same CIGARs, positions and HP as the CIGAR-only file, NOT the Python writer's random bytes),
    written slab by slab so that tens of GB need a few hundred MB of memory.  info: receives inflated_bytes / n_blocks."""
    import ctypes as C

    wl = synth.WORKLOADS[workload]
    if device is not None:
        d = synth.DeviceBatch(wl, device, 0, n_loci)
        cigar = d.cigar.cpu().numpy().view(np.uint32)
        reads = d.reads.cpu().numpy().view(np.uint8).reshape(-1).view(synth.READ_DTYPE)
        ls, le = d.locus_start.cpu().numpy(), d.locus_end.cpu().numpy()
        del d
    else:
        b = synth.generate_numpy(wl, 0, n_loci)
        cigar, reads, ls, le = b.cigar, b.reads, b.locus_start, b.locus_end
    n = len(reads)
    R = wl.reads_per_locus
    tid = (np.arange(n, dtype=np.int64) // R // LOCI_PER_CONTIG).astype(np.int32)
    key = tid.astype(np.int64) << 32 | reads["pos"].astype(np.int64)
    order = np.argsort(key, kind="stable").astype(np.uint64)
    name_id = np.arange(n, dtype=np.uint64)
    n_contigs = (n_loci + LOCI_PER_CONTIG - 1) // LOCI_PER_CONTIG
    with open(prefix + ".bed", "w") as bed:
        bed.write("".join(f"chr{j // LOCI_PER_CONTIG + 1}\t{s}\t{e}\n" for j, (s, e) in enumerate(zip(ls.tolist(), le.tolist()))))
    cigar = np.ascontiguousarray(cigar)
    reads = np.ascontiguousarray(reads)
    err = C.create_string_buffer(512)
    if seq:
        infl, nblk = C.c_uint64(0), C.c_uint64(0)
        rc = native_lib().inq_synth_write_bam_seq((prefix + ".bam").encode(), n, reads.ctypes.data, cigar.ctypes.data, order.ctypes.data,
                                                  tid.ctypes.data, name_id.ctypes.data, n_contigs, CONTIG_LEN, level,
                                                  threads or len(os.sched_getaffinity(0)), qual_mode, seed, slab_blocks,
                                                  C.byref(infl), C.byref(nblk), err, len(err))
        if info is not None:
            info.update(inflated_bytes=int(infl.value), n_blocks=int(nblk.value))
    else:
        rc = native_lib().inq_synth_write_bam((prefix + ".bam").encode(), n, reads.ctypes.data, cigar.ctypes.data, order.ctypes.data,
                                              tid.ctypes.data, name_id.ctypes.data, n_contigs, CONTIG_LEN, level,
                                              threads or len(os.sched_getaffinity(0)), err, len(err))
    if rc != 0:
        raise RuntimeError("synth_bam_writer: " + err.value.decode())
    return n


if __name__ == "__main__":
    if len(sys.argv) > 4 and sys.argv[4] in ("native", "native-seq", "native-seq-random"):  # native writer, workload generated on the GPU when there is one
        import torch

        n = write_native(sys.argv[1], int(sys.argv[2]), sys.argv[3], device=torch.device("cuda:0") if torch.cuda.is_available() else None,
                         seq=sys.argv[4] != "native", qual_mode=0 if sys.argv[4].endswith("random") else 1,
                         level=int(sys.argv[5]) if len(sys.argv) > 5 else 1, threads=int(sys.argv[6]) if len(sys.argv) > 6 else 0)
    else:
        n = write(sys.argv[1], int(sys.argv[2]), sys.argv[3], seq=len(sys.argv) > 4 and sys.argv[4] == "seq")
    print(f"wrote {sys.argv[3]}.bam/.bai/.bed: {n} reads, {os.path.getsize(sys.argv[3] + '.bam') / 1e6:.1f} MB")
