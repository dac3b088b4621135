"""ctypes binding of the image's libdeflate (runtime library only, 1.10: /usr/lib/x86_64-linux-gnu/libdeflate.so.0) - test and benchmark
plumbing, never the product.  Why: htslib is commonly built with libdeflate, so many real BAMs hold DEFLATE streams that zlib's
compressor would never write (other block splitting, other match choices - lazy and near-optimal parsing from level 8 on -, other code
shapes).  The device inflate (csrc/bgzf_inflate_wg.hip) is checked against such streams too, and against libdeflate's own decoder."""
from __future__ import annotations

import ctypes as C
import ctypes.util
import struct
import zlib

_lib = None


def available() -> bool:
    try:
        load()
        return True
    except OSError:
        return False


def load():
    global _lib
    if _lib is None:
        name = ctypes.util.find_library("deflate") or "libdeflate.so.0"
        L = C.CDLL(name)
        L.libdeflate_alloc_compressor.restype = C.c_void_p
        L.libdeflate_alloc_compressor.argtypes = [C.c_int]
        L.libdeflate_deflate_compress.restype = C.c_size_t
        L.libdeflate_deflate_compress.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.libdeflate_deflate_compress_bound.restype = C.c_size_t
        L.libdeflate_deflate_compress_bound.argtypes = [C.c_void_p, C.c_size_t]
        L.libdeflate_free_compressor.restype = None
        L.libdeflate_free_compressor.argtypes = [C.c_void_p]
        L.libdeflate_alloc_decompressor.restype = C.c_void_p
        L.libdeflate_alloc_decompressor.argtypes = []
        L.libdeflate_deflate_decompress.restype = C.c_int
        L.libdeflate_deflate_decompress.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.libdeflate_free_decompressor.restype = None
        L.libdeflate_free_decompressor.argtypes = [C.c_void_p]
        _lib = L
    return _lib


class Compressor:
    """Raw DEFLATE (RFC 1951) at libdeflate's level 0 - 12."""

    def __init__(self, level: int = 6):
        self._L = load()
        self._h = self._L.libdeflate_alloc_compressor(level)
        if not self._h:
            raise ValueError(f"libdeflate: no compressor at level {level}")

    def compress(self, data: bytes) -> bytes:
        cap = self._L.libdeflate_deflate_compress_bound(self._h, len(data))
        out = C.create_string_buffer(cap)
        n = self._L.libdeflate_deflate_compress(self._h, data, len(data), out, cap)
        if n == 0:
            raise RuntimeError("libdeflate_deflate_compress failed")
        return out.raw[:n]

    def close(self):
        if self._h:
            self._L.libdeflate_free_compressor(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def decompress(comp: bytes, out_len: int):
    """libdeflate's own decoder: (status, bytes) - status 0 = ok, 1 = bad data, 2 = short output, 3 = insufficient space."""
    L = load()
    d = L.libdeflate_alloc_decompressor()
    out = C.create_string_buffer(max(out_len, 1))
    got = C.c_size_t(0)
    rc = L.libdeflate_deflate_decompress(d, comp, len(comp), out, out_len, C.byref(got))
    L.libdeflate_free_decompressor(d)
    return rc, out.raw[: got.value]


def bgzf_block(data: bytes, level: int = 6, comp: "Compressor | None" = None) -> bytes:
    """One BGZF block (SAM spec 4.1) whose payload libdeflate compressed; AssertionError if it does not fit 64 KB."""
    own = comp is None
    c = comp or Compressor(level)
    payload = c.compress(data)
    if own:
        c.close()
    bsize = len(payload) + 25
    assert bsize <= 65535 and len(data) <= 65536
    hdr = struct.pack("<BBBBIBBHBBHH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 6, ord("B"), ord("C"), 2, bsize)
    return hdr + payload + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))
