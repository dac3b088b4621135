"""Kernel time of the BGZF inflate on BAM-like data: blocks of synthetic records (tools/make_synth_bam)
compressed like htslib does (zlib level 1..6, 0xff00 bytes per block)."""
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from inquistr_amd import hipcall, synth  # noqa: E402
from tools import bamio, make_synth_bam  # noqa: E402


def _block_writer():
    """bamio.bgzf_block (zlib), or CODEC=libdeflate: the same blocks as an htslib built with libdeflate writes them."""
    import os

    if os.environ.get("CODEC") == "libdeflate":
        from tools import libdeflate_shim

        comps = {}

        def mk(data, level):
            if level not in comps:
                comps[level] = libdeflate_shim.Compressor(level)
            return libdeflate_shim.bgzf_block(data, level, comps[level])

        return mk
    return bamio.bgzf_block


def make_blocks(n_blocks: int, level: int, kind: str = "cigar"):
    bgzf_block = _block_writer()
    if kind == "qual":  # what most of a real long-read BAM is: base qualities (Phred 0..50) and packed bases
        rng = np.random.default_rng(3)
        base = (rng.integers(0, 51, 64 * bamio.BLOCK, dtype=np.uint8)).tobytes()
        comp = [bgzf_block(base[i : i + bamio.BLOCK], level) for i in range(0, len(base), bamio.BLOCK)]
        return b"".join((comp * (n_blocks // len(comp) + 1))[:n_blocks])
    if kind in ("seq", "ont"):
        # seq: packed bases (two per byte, 16 equiprobable byte values); ont: long reads as a nanopore BAM holds them:
        # 12 KB of packed bases, then 24 KB of base qualities (skewed Phred), a few tags
        rng = np.random.default_rng(4)
        nib = np.array([1, 2, 4, 8], dtype=np.uint8)
        n = 48 * bamio.BLOCK
        packed = (nib[rng.integers(0, 4, n)] << 4 | nib[rng.integers(0, 4, n)]).astype(np.uint8)
        if kind == "seq":
            base = packed.tobytes()
        else:
            q = np.clip(rng.gamma(4.0, 5.0, n), 1, 50).astype(np.uint8)
            parts, at = [], 0
            while at + 36_000 < n:
                parts += [bytes(36), packed[at : at + 12_000].tobytes(), q[at : at + 24_000].tobytes(), b"MLB" + q[at : at + 400].tobytes()]
                at += 36_000
            base = b"".join(parts)
        comp = [bgzf_block(base[i : i + bamio.BLOCK], level) for i in range(0, len(base) - bamio.BLOCK, bamio.BLOCK)]
        return b"".join((comp * (n_blocks // len(comp) + 1))[:n_blocks])
    wl = synth.WORKLOADS["unphased100k"]
    need = n_blocks * bamio.BLOCK
    blob = bytearray()
    g = 0
    while len(blob) < need and g < 4000:
        b = synth.generate_numpy(wl, g, g + 200)
        data, *_ = make_synth_bam.records_for(b, 0, g * 30)
        blob += data
        g += 200
    base = bytes(blob)
    comp = [bgzf_block(base[i : i + bamio.BLOCK], level) for i in range(0, len(base) - bamio.BLOCK, bamio.BLOCK)]
    comp = (comp * (n_blocks // len(comp) + 1))[:n_blocks]
    return b"".join(comp)


def main():
    n_blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    kind = sys.argv[3] if len(sys.argv) > 3 else "cigar"
    comp = make_blocks(n_blocks, level, kind)
    blocks = hipcall.scan_bgzf(comp)
    import os

    alt = os.environ.get("INQ_LIB")  # another build of libinquistr_hip.so (A/B)
    ctx = hipcall.Context(0, lib=hipcall.load(alt)) if alt else hipcall.Context(0)
    out_bytes = int(blocks["isize"].sum())
    if os.environ.get("ALGO"):
        ctx.set_option("inflate_algo", int(os.environ["ALGO"]))
    if os.environ.get("PAIRS"):
        ctx.set_option("inflate_lit_pairs", int(os.environ["PAIRS"]))
    if os.environ.get("TOKENS"):
        ctx.set_option("inflate_tokens", int(os.environ["TOKENS"]))
    if os.environ.get("NOCRC"):
        ctx.set_option("verify_crc", 0)
    for rep in range(3):
        t = time.perf_counter()
        rc, out, st = ctx.bgzf_inflate(comp, blocks, check=False)
        wall = time.perf_counter() - t
        ms, _ = ctx.timing_read(2)
        print(f"blocks {len(blocks)} level {level} {kind}: comp {len(comp) / 1e6:.1f} MB -> {out_bytes / 1e6:.1f} MB, kernel {ms:.2f} ms "
              f"= {out_bytes / ms / 1e6:.2f} GB/s out, {len(comp) / ms / 1e6:.2f} GB/s in (call {wall * 1e3:.0f} ms)", flush=True)
    import os

    if os.environ.get("INQ_INFLATE_DEBUG"):
        if int(os.environ["INQ_INFLATE_DEBUG"]) & 8:
            v = st[: len(st) // 8 * 8].reshape(-1, 8).astype(np.float64)
            first = st[: len(st) // 8 * 8].reshape(-1, 8)[:, 0].astype(np.int64)
            print(f"  stretches: mean {((first >> 8) & 0xfff).mean():.1f}  sweeps: mean {(first >> 20).mean():.1f}")
            v[:, 0] = first & 0xff
            second = st[: len(st) // 8 * 8].reshape(-1, 8)[:, 1].astype(np.int64)
            print(f"  kcyc in the sweeps (the rest of 'matches' is the gather): mean {(second >> 8).mean():.1f}")
            v[:, 1] = second & 0xff
            names = ["deflate blocks", "rounds", "count passes", "kcyc header parse", "kcyc tables", "kcyc counting", "kcyc commit", "kcyc matches"]
            for k, nm in enumerate(names):
                print(f"  {nm}: mean {v[:, k].mean():.1f} median {np.median(v[:, k]):.1f} max {v[:, k].max():.0f}")
        if int(os.environ["INQ_INFLATE_DEBUG"]) & 4:
            kc = st.astype(np.float64)
            print(f"shader kilo-cycles per lane: median {np.median(kc):.0f}, max {kc.max():.0f}; with the kernel time above that is "
                  f"{kc.max() * 1024 / (ms * 1e-3) / 1e9:.2f} GHz if the slowest lane spans the kernel")
        return
    # spot check against zlib
    b = blocks[len(blocks) // 2]
    want = zlib.decompressobj(-15).decompress(comp[int(b["comp_off"]) : int(b["comp_off"]) + int(b["comp_len"])])
    assert out[int(b["out_off"]) : int(b["out_off"]) + int(b["isize"])].tobytes() == want


if __name__ == "__main__":
    main()
