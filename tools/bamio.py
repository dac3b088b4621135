"""Minimal BAM + BAI writer (stdlib zlib + struct): test / benchmark plumbing, never the product.

Writes what the C++ front end reads: BGZF blocks (records may span blocks), header with @SQ text and
binary references, records with CIGAR / HP / SA / CG tags, SEQ '*' unless asked otherwise, and a
.bai with bins, 16 kb linear index and the htslib metadata pseudo-bin.
"""
from __future__ import annotations

import struct
import zlib
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

OPS = "MIDNSHP=X"
BLOCK = 0xFF00

EOF_BLOCK = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def reg2bin(beg: int, end: int) -> int:
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def ref_span(cigar: Sequence[Tuple[str, int]]) -> int:
    return sum(n for o, n in cigar if o in "MDN=X")


def encode_aux(tags: Sequence[Tuple[str, str, object]]) -> bytes:
    out = b""
    for tag, typ, val in tags:
        out += tag.encode() + typ.encode()
        if typ == "A":
            out += str(val).encode()[:1]
        elif typ in "cCsSiI":
            out += struct.pack("<" + {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I"}[typ], val)
        elif typ == "f":
            out += struct.pack("<f", val)
        elif typ in "ZH":
            out += str(val).encode() + b"\0"
        elif typ == "B":
            sub, arr = val
            out += sub.encode() + struct.pack("<I", len(arr))
            out += struct.pack("<%d%s" % (len(arr), {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub]), *arr)
        else:
            raise ValueError(typ)
    return out


def encode_record(name: str, flag: int, tid: int, pos: int, mapq: int, cigar: Sequence[Tuple[str, int]],
                  tags: Sequence[Tuple[str, str, object]] = (), l_seq: int = 0) -> bytes:
    """One BAM record (without the leading block_size).  CIGARs beyond 65535 ops go to CG:B,I."""
    words = [(n << 4) | OPS.index(o) for o, n in cigar]
    tags = list(tags)
    end = pos + (ref_span(cigar) or 1 if not (flag & 4) else 1)
    if len(words) > 65535:
        tags.append(("CG", "B", ("I", words)))
        words = [(l_seq << 4) | 4, (ref_span(cigar) << 4) | 3]
    rn = name.encode() + b"\0"
    core = struct.pack("<iiBBHHHIiii", tid, pos, len(rn), mapq, reg2bin(max(pos, 0), max(end, 1)), len(words), flag, l_seq, -1, -1, 0)
    seq = b"\0" * ((l_seq + 1) // 2) + b"\xff" * l_seq
    return core + rn + struct.pack("<%dI" % len(words), *words) + seq + encode_aux(tags)


def bgzf_block(data: bytes, level: int = 1) -> bytes:
    comp = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = comp.compress(data) + comp.flush()
    crc = zlib.crc32(data) & 0xFFFFFFFF
    bsize = len(body) + 25
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize) + body + struct.pack("<II", crc, len(data)))


class BamWriter:
    def __init__(self, path: str, refs: Sequence[Tuple[str, int]], header_text: Optional[str] = None, level: int = 1,
                 block: int = BLOCK):
        self.path, self.refs, self.level, self.block = path, list(refs), level, block
        if header_text is None:
            header_text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in refs)
        h = b"BAM\1" + struct.pack("<I", len(header_text)) + header_text.encode() + struct.pack("<I", len(refs))
        for n, l in refs:
            nb = n.encode() + b"\0"
            h += struct.pack("<I", len(nb)) + nb + struct.pack("<I", l)
        self.buf = bytearray(h)
        self.recs: List[Tuple[int, int, int, int, int, int]] = []  # (ustart, uend, tid, beg, end, flag)

    def add(self, name, flag, tid, pos, mapq, cigar, tags=(), l_seq=0):
        body = encode_record(name, flag, tid, pos, mapq, cigar, tags, l_seq)
        u0 = len(self.buf)
        self.buf += struct.pack("<I", len(body)) + body
        end = pos + (ref_span(cigar) or 1 if not (flag & 4) else 1)
        self.recs.append((u0, len(self.buf), tid, pos, end, flag))

    def add_raw(self, block: bytes, tid: int, beg: int, end: int, flag: int = 0):
        """block = block_size + record bytes, already encoded (fast path of the synthetic generator)."""
        u0 = len(self.buf)
        self.buf += block
        self.recs.append((u0, len(self.buf), tid, beg, end, flag))

    def close(self, write_index: bool = True, index: str = "bai", csi_min_shift: int = 14, csi_depth: int = 5):
        """index: "bai", "csi" (BGZF-compressed CSI v1 with the given binning) or "both"."""
        data = bytes(self.buf)
        coff, out = [], bytearray()
        for i in range(0, max(len(data), 1), self.block):
            coff.append(len(out))
            out += bgzf_block(data[i : i + self.block], self.level)
        end_coff = len(out)
        out += EOF_BLOCK
        with open(self.path, "wb") as f:
            f.write(out)

        def vo(u):
            blk, within = divmod(u, self.block)
            if blk >= len(coff):
                return end_coff << 16
            return (coff[blk] << 16) | within

        if write_index and index in ("bai", "both"):
            self._write_bai(vo)
        if write_index and index in ("csi", "both"):
            self._write_csi(vo, csi_min_shift, csi_depth)

    def _write_csi(self, vo, min_shift: int, depth: int):
        """[3P] htslib CSI v1: per bin the chunk list plus loffset = linear-index entry of the bin's first window, where the
        linear index (one entry per 2^min_shift window, first record overlapping it) has its empty windows filled from the
        NEXT filled one; metadata pseudo-bin as in a .bai; the whole file BGZF-compressed."""
        def reg2bin(beg, end):
            end -= 1
            s, t = min_shift, ((1 << depth * 3) - 1) // 7
            for l in range(depth, 0, -1):
                if beg >> s == end >> s:
                    return t + (beg >> s)
                s += 3
                t -= 1 << (l - 1) * 3
            return 0

        def bin_first_window(b):
            l = 0
            while b >= ((1 << 3 * (l + 1)) - 1) // 7:
                l += 1
            return (b - ((1 << 3 * l) - 1) // 7) << (3 * (depth - l))

        nref = len(self.refs)
        bins = [dict() for _ in range(nref)]
        lin = [dict() for _ in range(nref)]
        meta = [[None, None, 0, 0] for _ in range(nref)]
        n_no_coor = 0
        for u0, u1, tid, beg, end, flag in self.recs:
            if tid < 0:
                n_no_coor += 1
                continue
            v0, v1 = vo(u0), vo(u1)
            ch = bins[tid].setdefault(reg2bin(max(beg, 0), max(end, 1)), [])
            if ch and ch[-1][1] == v0:
                ch[-1][1] = v1
            else:
                ch.append([v0, v1])
            for w in range(max(beg, 0) >> min_shift, ((max(end, 1) - 1) >> min_shift) + 1):
                lin[tid].setdefault(w, v0)
            m = meta[tid]
            m[0] = v0 if m[0] is None else min(m[0], v0)
            m[1] = v1 if m[1] is None else max(m[1], v1)
            m[3 if flag & 4 else 2] += 1
        out = bytearray(b"CSI\1" + struct.pack("<iii", min_shift, depth, 0) + struct.pack("<i", nref))
        meta_bin = ((1 << (depth + 1) * 3) - 1) // 7 + 1
        for t in range(nref):
            nwin = (max(lin[t]) + 1) if lin[t] else 0
            filled, nxt = [0] * nwin, 0
            for w in range(nwin - 1, -1, -1):
                nxt = lin[t].get(w, nxt)
                filled[w] = nxt
            out += struct.pack("<i", len(bins[t]) + (1 if meta[t][0] is not None else 0))
            for b in sorted(bins[t]):
                w0 = bin_first_window(b)
                out += struct.pack("<IQi", b, filled[w0] if w0 < nwin else 0, len(bins[t][b]))
                for v0, v1 in bins[t][b]:
                    out += struct.pack("<QQ", v0, v1)
            if meta[t][0] is not None:
                out += struct.pack("<IQi", meta_bin, 0, 2) + struct.pack("<QQQQ", meta[t][0], meta[t][1], meta[t][2], meta[t][3])
        out += struct.pack("<Q", n_no_coor)
        data = bytes(out)
        with open(self.path + ".csi", "wb") as f:
            for i in range(0, len(data), self.block):
                f.write(bgzf_block(data[i : i + self.block], 6))
            f.write(EOF_BLOCK)

    def _write_bai(self, vo):
        nref = len(self.refs)
        bins: List[Dict[int, List[List[int]]]] = [dict() for _ in range(nref)]
        lin: List[Dict[int, int]] = [dict() for _ in range(nref)]
        meta = [[None, None, 0, 0] for _ in range(nref)]
        n_no_coor = 0
        for u0, u1, tid, beg, end, flag in self.recs:
            if tid < 0:
                n_no_coor += 1
                continue
            v0, v1 = vo(u0), vo(u1)
            b = reg2bin(max(beg, 0), max(end, 1))
            ch = bins[tid].setdefault(b, [])
            if ch and ch[-1][1] == v0:
                ch[-1][1] = v1
            else:
                ch.append([v0, v1])
            for w in range(max(beg, 0) >> 14, ((max(end, 1) - 1) >> 14) + 1):
                if w not in lin[tid]:
                    lin[tid][w] = v0
            m = meta[tid]
            m[0] = v0 if m[0] is None else min(m[0], v0)
            m[1] = v1 if m[1] is None else max(m[1], v1)
            if flag & 4:
                m[3] += 1
            else:
                m[2] += 1
        out = bytearray(b"BAI\1" + struct.pack("<I", nref))
        for t in range(nref):
            nb = len(bins[t]) + (1 if meta[t][0] is not None else 0)
            out += struct.pack("<I", nb)
            for b in sorted(bins[t]):
                out += struct.pack("<II", b, len(bins[t][b]))
                for v0, v1 in bins[t][b]:
                    out += struct.pack("<QQ", v0, v1)
            if meta[t][0] is not None:
                out += struct.pack("<II", 37450, 2) + struct.pack("<QQQQ", meta[t][0], meta[t][1], meta[t][2], meta[t][3])
            nint = (max(lin[t]) + 1) if lin[t] else 0
            out += struct.pack("<I", nint)
            # [3P] htslib (hts_idx_finish) fills a window no record overlaps with the offset of the NEXT filled window, from
            # the right: the reference's own small-test.bam.bai carries the first record's offset in all 9 326 windows in
            # front of it (tests/test_reference_bai_pin.py).  (samtools 0.1's fill_missing copied the previous one.)
            filled = [0] * nint
            nxt = 0
            for w in range(nint - 1, -1, -1):
                if w in lin[t]:
                    nxt = lin[t][w]
                filled[w] = nxt
            out += struct.pack("<%dQ" % nint, *filled)
        out += struct.pack("<Q", n_no_coor)
        with open(self.path + ".bai", "wb") as f:
            f.write(out)


def inflate_span(comp, blocks) -> bytes:
    """zlib-inflates the blocks of a span (block table as inq_bgzf_block_t records) into one byte string."""
    mv = memoryview(comp)
    out = []
    for b in blocks:
        o, n = int(b["comp_off"]), int(b["comp_len"])
        d = zlib.decompressobj(-15).decompress(bytes(mv[o : o + n]))
        assert len(d) == int(b["isize"])
        out.append(d)
    return b"".join(out)


_AUX_FIXED = {"A": 1, "c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4, "d": 8}
_AUX_FMT = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}


def read_records(u: bytes, off: int):
    """Walks BAM records in inflated bytes from `off` until one is cut by the end.  Yields dicts with the
    fields the path reads: off, tid, pos, mapq, flag, cigar (words, CG-swapped), hp (type, value) / None,
    sa (type, value) / None."""
    n = len(u)
    while off + 4 <= n:
        (bs,) = struct.unpack_from("<I", u, off)
        if bs < 32 or off + 4 + bs > n:
            return
        b = off + 4
        tid, pos, l_rn, mapq, _bin, n_cig, flag, l_seq = struct.unpack_from("<iiBBHHHI", u, b)
        p = b + 32 + l_rn
        words = list(struct.unpack_from("<%dI" % n_cig, u, p))
        p += 4 * n_cig + (l_seq + 1) // 2 + l_seq
        end = b + bs
        hp = sa = cg = None
        while p + 3 <= end:
            tag, typ = u[p : p + 2].decode("latin1"), chr(u[p + 2])
            v = p + 3
            if typ in _AUX_FIXED:
                sz = _AUX_FIXED[typ]
                val = u[v : v + 1].decode("latin1") if typ == "A" else (struct.unpack_from("<" + _AUX_FMT[typ], u, v)[0] if typ != "d" else None)
            elif typ in "ZH":
                z = u.index(b"\0", v, end) if b"\0" in u[v:end] else -1
                if z < 0:
                    break
                sz, val = z - v + 1, u[v:z].decode("latin1")
            elif typ == "B":
                sub = chr(u[v])
                (cnt,) = struct.unpack_from("<I", u, v + 1)
                es = 1 if sub in "cC" else 2 if sub in "sS" else 4
                sz = 5 + cnt * es
                val = (sub, list(struct.unpack_from("<%d%s" % (cnt, _AUX_FMT[sub]), u, v + 5))) if v + sz <= end else None
            else:
                break
            if v + sz > end:
                break
            if tag == "HP" and hp is None:
                hp = (typ, val)
            elif tag == "SA" and sa is None:
                sa = (typ, val)
            elif tag == "CG" and cg is None:
                cg = (typ, val)
            p = v + sz
        if cg and cg[0] == "B" and cg[1][0] in "Ii" and words and pos >= 0 and (words[0] & 15) == 4 and (words[0] >> 4) == l_seq \
                and len(cg[1][1]) >= len(words):
            words = [w & 0xFFFFFFFF for w in cg[1][1]]
        yield dict(off=off, next=b + bs, tid=tid, pos=pos, mapq=mapq, flag=flag, cigar=words, hp=hp, sa=sa)
        off = b + bs
