"""outlier_oracle — naive restatement of `inquiSTR outlier`, the consumer of combined .inq files.

TEST INFRASTRUCTURE ONLY.  Pure Python with numpy float32 scalars so that every f32 rounding of the
reference happens here too.  Reference: wdecoster/inquiSTR v0.13.0, src/outlier.rs; lines cited per function.

Pinned by the reference's own unit tests `test_z_score_outliers` / `test_dbscan_outliers`
(src/outlier.rs:148-168; both expect ["s11"]), kept in tests/golden/kat_outlier.json.  The DBSCAN model is
the third-party crate `dbscan` 0.3.1 ([3P], not vendored): restated from its published algorithm —
neighbours are the points at euclidean distance < eps (strict, the point itself included), a point with at
least `min_points` neighbours is a core point, and what is neither a core point nor a neighbour of one
stays `Noise`; which cluster an edge point lands in depends on visiting order, whether a point is noise
does not.  `mode` (src/outlier.rs:132-145) breaks ties between equally frequent values by HashMap
iteration order, which Rust randomises per process: the reference itself is ambiguous there; this
restatement (and the kernel) take the smallest of the tied values.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import numpy as np

F = np.float32


class ReferencePanic(Exception):
    pass


_F32_RE = __import__("re").compile(r"^[+-]?(?:inf|infinity|nan|(?:\d+\.?\d*|\.\d+)(?:e[+-]?\d+)?)$", __import__("re").IGNORECASE)


def parse_f32(s: str) -> np.float32:
    """Rust `str::parse::<f32>` as used by get_repeat_lengths (src/outlier.rs:78): optional sign, then
    inf / infinity / nan (any case) or digits with an optional point and exponent; nothing else."""
    if not _F32_RE.match(s):
        raise ReferencePanic("Failed to parse number")
    return F(np.float32(s))  # correctly rounded, like Rust's


def std_deviation_and_mean(data: Sequence[np.float32]):
    """src/outlier.rs:18-31: sequential f32 sums."""
    with np.errstate(all="ignore"):  # inf - inf and friends are part of the semantics, not accidents
        s = F(0)
        for v in data:
            s = F(s + v)
        count = F(len(data))
        mean = F(s / count)
        var = F(0)
        for v in data:
            d = F(mean - v)
            var = F(var + F(d * d))
        var = F(var / count)
        return mean, F(np.sqrt(var))


def get_repeat_lengths(fields: Sequence[str], minsize: int) -> Optional[List[np.float32]]:
    """src/outlier.rs:75-95"""
    values = [parse_f32(x) for x in fields[3:]]
    values = [F(0) if np.isnan(v) else v for v in values]
    if not values:
        raise ReferencePanic("called `Option::unwrap()` on a `None` value")
    if max(values) < F(minsize):
        return None
    return values


def _strip(name: str) -> str:
    return name.replace("_H1", "").replace("_H2", "")


def z_score_flags(values: Sequence[np.float32], cutoff: float) -> List[bool]:
    """src/outlier.rs:97-110, as per-value flags"""
    mean, sd = std_deviation_and_mean(values)
    with np.errstate(divide="ignore", invalid="ignore"):
        return [bool(F(F(v - mean) / sd) >= F(cutoff)) for v in values]


def mode(values: Sequence[np.float32]) -> int:
    """src/outlier.rs:132-145 (ties: smallest value, see the header)"""
    counts = {}
    for v in values:
        if v > 0:
            k = int(min(float(v), 2.0**64 - 1))  # `as usize` saturates
            counts[k] = counts.get(k, 0) + 1
    if not counts:
        raise ReferencePanic("No mode found for repeat")
    best = max(counts.values())
    return min(k for k, c in counts.items() if c == best)


def dbscan_flags(values: Sequence[np.float32], mincluster: int) -> List[bool]:
    """src/outlier.rs:112-130 + [3P] dbscan 0.3.1 Model::run, as per-value "is noise" flags"""
    eps = float(max(2 * mode(values), 10))
    v = [float(x) for x in values]
    n = len(v)
    core = [sum(1 for j in range(n) if abs(v[i] - v[j]) < eps) >= mincluster for i in range(n)]
    return [not core[i] and not any(core[j] and abs(v[i] - v[j]) < eps for j in range(n)) for i in range(n)]


def z_score_outliers(values, samples, cutoff):
    return [_strip(samples[i]) for i, f in enumerate(z_score_flags(values, cutoff)) if f]


def dbscan_outliers(values, samples, mincluster):
    return [_strip(samples[i]) for i, f in enumerate(dbscan_flags(values, mincluster)) if f]


def outlier_text(lines: Sequence[str], minsize: int = 10, zscore_cutoff: float = 3.0, method: str = "zscore",
                 subset: Optional[Sequence[str]] = None) -> str:
    """src/outlier.rs:33-73 on the lines of a combined file (without their newlines)."""
    if not lines:
        raise ReferencePanic("called `Option::unwrap()` on a `None` value")
    out = ["chrom\tbegin\tend\toutliers"]
    samples = lines[0].split("\t")[3:]
    if not samples:
        raise ReferencePanic("argument of integer logarithm must be positive")
    mincluster = len(samples).bit_length() - 1  # usize::ilog2
    for line in lines[1:]:
        f = line.split("\t")
        if len(f) < 3:
            raise ReferencePanic("index out of bounds")
        values = get_repeat_lengths(f, minsize)
        if values is None:
            continue
        flags = z_score_flags(values, zscore_cutoff) if method == "zscore" else dbscan_flags(values, mincluster)
        hit = [i for i, x in enumerate(flags) if x]
        if any(i >= len(samples) for i in hit):
            raise ReferencePanic("index out of bounds")
        names = [_strip(samples[i]) for i in hit]
        if names and (subset is None or any(n in subset for n in names)):
            out.append(f"{f[0]}\t{f[1]}\t{f[2]}\t{','.join(names)}")
    return "\n".join(out) + "\n"


def c_outlier_rows(values: np.ndarray, row_len: np.ndarray, method: str = "zscore", minsize: int = 10, cutoff: float = 3.0,
                   mincluster: int = 1, threads: int = 1):
    """The C restatement (oracle/outlier_oracle.c, OpenMP over rows): (flags, keep).  CPU baseline + cross-check."""
    import ctypes as C
    import os
    import subprocess

    here = os.path.dirname(os.path.abspath(__file__))
    lib = os.path.join(here, "liborcoutlier.so")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(os.path.join(here, "outlier_oracle.c")):
        subprocess.check_call(["make", "-C", here, "-B", "liborcoutlier.so"], stdout=subprocess.DEVNULL)
    L = C.CDLL(lib)
    values = np.ascontiguousarray(values, dtype=np.float32)
    row_len = np.ascontiguousarray(row_len, dtype=np.uint32)
    n_rows, stride = values.shape
    flags = np.zeros((n_rows, stride), dtype=np.uint8)
    keep = np.zeros(n_rows, dtype=np.uint8)
    L.orc_outlier_rows.restype = None
    L.orc_outlier_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_uint32, C.c_float, C.c_uint32,
                                   C.c_void_p, C.c_void_p, C.c_int]
    L.orc_outlier_rows(values.ctypes.data, row_len.ctypes.data, n_rows, stride, {"zscore": 0, "dbscan": 1}[method], minsize,
                       cutoff, mincluster, flags.ctypes.data, keep.ctypes.data, threads)
    return flags, keep
