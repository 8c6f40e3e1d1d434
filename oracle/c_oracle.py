"""ctypes loader for oracle/cco_oracle.c (CPU ORACLE -- test infrastructure, not product code).

Allowed importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liburcco_oracle.so")
_lib = None

i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


RNG_MIX32 = 0x100   # OR-ed into row_rate_mode: the 32-bit form of the down-sampling RNG (cco_oracle.c ORC_RNG_MIX32)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "cco_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_llr_k.restype = C.c_double
        L.orc_llr_k.argtypes = [C.c_int64] * 4
        L.orc_llr.restype = C.c_double
        L.orc_llr.argtypes = [C.c_int64] * 4
        L.orc_u01.restype = C.c_double
        L.orc_u01.argtypes = [C.c_uint32] * 3
        L.orc_u01_mix32.restype = C.c_double
        L.orc_u01_mix32.argtypes = [C.c_uint32] * 3
        L.orc_mix32.restype = C.c_uint32
        L.orc_mix32.argtypes = [C.c_uint32] * 3
        L.orc_column_counts.restype = None
        L.orc_column_counts.argtypes = [C.c_int64, i32p, C.c_int32, i32p]
        L.orc_downsample.restype = C.c_int64
        L.orc_downsample.argtypes = [C.c_int64, i64p, i32p, i32p, C.c_uint32, C.c_int32, C.c_int, C.c_int64, i64p, i32p]
        L.orc_transpose.restype = None
        L.orc_transpose.argtypes = [C.c_int64, i64p, i32p, C.c_int32, i64p, i32p]
        L.orc_cco_rows.restype = C.c_int64
        L.orc_cco_rows.argtypes = [C.c_int32, C.c_int32, i64p, i32p, i64p, i32p, C.c_int32, i32p, i32p, C.c_int64, C.c_int,
                                   C.c_int32, C.c_int, C.c_double, i32p, i32p, f64p, C.c_int]
        L.orc_max_threads.restype = C.c_int
        _lib = L
    return _lib


class Csr:
    """Binary CSR: row_ptr int64[n_rows+1], col_idx int32 (sorted, unique per row)."""

    def __init__(self, n_rows: int, n_cols: int, row_ptr: np.ndarray, col_idx: np.ndarray):
        self.n_rows, self.n_cols = int(n_rows), int(n_cols)
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
        self.col_idx = np.ascontiguousarray(col_idx, dtype=np.int32)
        assert self.row_ptr.shape == (self.n_rows + 1,)

    @property
    def nnz(self) -> int:
        return int(self.row_ptr[-1])

    @staticmethod
    def from_rows(rows: Sequence[Sequence[int]], n_cols: int) -> "Csr":
        rp = np.zeros(len(rows) + 1, np.int64)
        rp[1:] = np.cumsum([len(r) for r in rows])
        ci = np.fromiter((j for r in rows for j in r), np.int32, count=int(rp[-1]))
        return Csr(len(rows), n_cols, rp, ci)


def column_counts(m: Csr) -> np.ndarray:
    out = np.zeros(max(m.n_cols, 1), np.int32)
    lib().orc_column_counts(m.nnz, m.col_idx if m.nnz else np.zeros(1, np.int32), m.n_cols, out)
    return out[:m.n_cols]


def downsample(m: Csr, raw_counts: np.ndarray, seed: int, max_n: int, row_rate_mode: int = 0, row_base: int = 0) -> Csr:
    out_rp = np.zeros(m.n_rows + 1, np.int64)
    out_ci = np.zeros(max(m.nnz, 1), np.int32)
    rc = np.ascontiguousarray(raw_counts, np.int32)
    if rc.size == 0:
        rc = np.zeros(1, np.int32)
    nnz = lib().orc_downsample(m.n_rows, m.row_ptr, m.col_idx if m.nnz else np.zeros(1, np.int32), rc,
                               seed & 0xFFFFFFFF, max_n, row_rate_mode, row_base, out_rp, out_ci)
    return Csr(m.n_rows, m.n_cols, out_rp, out_ci[:nnz].copy())


def transpose(m: Csr) -> Tuple[np.ndarray, np.ndarray]:
    col_ptr = np.zeros(m.n_cols + 1, np.int64)
    row_idx = np.zeros(max(m.nnz, 1), np.int32)
    lib().orc_transpose(m.n_rows, m.row_ptr, m.col_idx if m.nnz else np.zeros(1, np.int32), m.n_cols, col_ptr, row_idx)
    return col_ptr, row_idx[:m.nnz]


class IndicatorRows:
    """Strided top-k output for item rows [item_lo, item_hi)."""

    def __init__(self, item_lo, item_hi, k, count, idx, llr, pairs):
        self.item_lo, self.item_hi, self.k = item_lo, item_hi, k
        self.count, self.idx, self.llr, self.pairs = count, idx, llr, pairs

    def row(self, i: int) -> List[Tuple[int, float]]:
        r = i - self.item_lo
        c = int(self.count[r])
        return list(zip(self.idx[r, :c].tolist(), self.llr[r, :c].tolist()))

    def to_csr(self):
        rp = np.zeros(self.count.size + 1, np.int64)
        rp[1:] = np.cumsum(self.count)
        mask = np.arange(self.k)[None, :] < self.count[:, None]
        return rp, self.idx[mask], self.llr[mask]


def cco_rows(a_col_ptr, a_row_idx, b: Csr, cnt_a, cnt_b, n_users: int, exclude_self: bool, k: int,
             min_llr: Optional[float] = None, item_lo: int = 0, item_hi: Optional[int] = None, n_threads: int = 1) -> IndicatorRows:
    n_items_a = a_col_ptr.size - 1
    if item_hi is None:
        item_hi = n_items_a
    n = item_hi - item_lo
    count = np.zeros(max(n, 1), np.int32)
    idx = np.zeros((max(n, 1), k), np.int32)
    llr = np.zeros((max(n, 1), k), np.float64)
    one32 = np.zeros(1, np.int32)
    pairs = lib().orc_cco_rows(item_lo, item_hi, np.ascontiguousarray(a_col_ptr, np.int64),
                               np.ascontiguousarray(a_row_idx, np.int32) if a_row_idx.size else one32,
                               b.row_ptr, b.col_idx if b.nnz else one32, b.n_cols,
                               np.ascontiguousarray(cnt_a, np.int32) if len(cnt_a) else one32,
                               np.ascontiguousarray(cnt_b, np.int32) if len(cnt_b) else one32,
                               n_users, int(exclude_self), k, int(min_llr is not None),
                               float(min_llr) if min_llr is not None else 0.0, count, idx, llr, n_threads)
    return IndicatorRows(item_lo, item_hi, k, count[:n], idx[:n], llr[:n], int(pairs))


class DatasetParams:
    def __init__(self, max_elements_per_row: int = 500, max_interesting_elements: int = 50, min_llr: Optional[float] = None):
        self.max_elements_per_row = max_elements_per_row
        self.max_interesting_elements = max_interesting_elements
        self.min_llr = min_llr


def cross_occurrence_downsampled(mats: Sequence[Csr], params: Sequence[DatasetParams], seed: int, row_rate_mode: int = 0,
                                 n_threads: int = 1, item_lo: int = 0, item_hi: Optional[int] = None) -> List[IndicatorRows]:
    """Whole path on CSR inputs (mats[0] = primary): SimilarityAnalysis.crossOccurrenceDownsampled."""
    a_raw = mats[0]
    a = downsample(a_raw, column_counts(a_raw), seed, params[0].max_elements_per_row, row_rate_mode)
    cnt_a = column_counts(a)
    a_cp, a_ri = transpose(a)
    out = []
    for d, (m, p) in enumerate(zip(mats, params)):
        if d == 0:
            b, cnt_b = a, cnt_a
        else:
            assert m.n_rows == a_raw.n_rows
            b = downsample(m, column_counts(m), seed, p.max_elements_per_row, row_rate_mode)
            cnt_b = column_counts(b)
        out.append(cco_rows(a_cp, a_ri, b, cnt_a, cnt_b, a_raw.n_rows, d == 0, p.max_interesting_elements, p.min_llr,
                            item_lo, item_hi, n_threads))
    return out
