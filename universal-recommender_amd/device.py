"""Device-level driver of liburcco: the CCO model build with every matrix resident in HBM.

PyTorch is plumbing here (device memory, the HIP stream, torch.distributed for the RCCL exchange in
sharded.py); all arithmetic is the hand-written HIP behind include/urcco.h.  The stage order is the one Mahout's
SimilarityAnalysis.crossOccurrenceDownsampled runs on Spark (reference call sites URAlgorithm.scala:323-346):
column counts -> sampleDownAndBinarize -> column counts of the sample -> A.t -> per event type: A.t %*% B fused with
computeSimilarities (LLR + top-k).  Nothing in this path synchronises with the host.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import torch

from . import _lib


@dataclass
class DatasetParams:
    """Mahout DownsamplableCrossOccurrenceDataset limits, as URAlgorithm.scala:334-341 fills them."""
    max_elements_per_row: int = 500          # indicators[i].maxItemsPerUser / maxEventsPerEventType
    max_interesting_elements: int = 50       # indicators[i].maxCorrelatorsPerItem / maxCorrelatorsPerEventType
    min_llr: Optional[float] = None          # indicators[i].minLLR


@dataclass
class DevCsr:
    """Binary user x item matrix in HBM.  col_idx may be longer than the live nnz (capacity)."""
    n_rows: int
    n_cols: int
    row_ptr: torch.Tensor   # int64 [n_rows + 1]
    col_idx: torch.Tensor   # int32 [>= nnz]
    nnz_bound: int          # upper bound on nnz known to the host without a sync


@dataclass
class DevIndicators:
    """One indicator matrix (rows = items of A in [item_lo, item_hi), cols = items of B_i), CSR in HBM."""
    item_lo: int
    item_hi: int
    n_cols: int
    k: int
    row_ptr: torch.Tensor   # int64 [n + 1]
    col_idx: torch.Tensor   # int32 [n * k] (first row_ptr[-1] live)
    llr: torch.Tensor       # float64 [n * k]
    stats: torch.Tensor     # int64 [STATS_LEN]: pairs, then rows / pairs / users / emitted entries per accumulator bin
    sampled_row_ptr: Optional[torch.Tensor] = None  # row_ptr of the down-sampled B (its last entry = nnz')
    sampled_col_idx: Optional[torch.Tensor] = None  # col_idx of the down-sampled B (first nnz' entries live)

    def to_host(self):
        rp = self.row_ptr.cpu().numpy()
        nnz = int(rp[-1])
        return rp, self.col_idx[:nnz].cpu().numpy(), self.llr[:nnz].cpu().numpy()


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class DeviceSession:
    """urcco_session bound to a torch device; launches go on torch's current stream for that device."""

    def __init__(self, device: torch.device, library=None, stream: Optional["torch.cuda.Stream"] = None):
        self.device = torch.device(device)
        self.lib = library if library is not None else _lib.lib()
        self.torch_stream = None
        handle = C.c_void_p()
        if self.device.type == "cuda":
            index = self.device.index if self.device.index is not None else torch.cuda.current_device()
            self.device = torch.device("cuda", index)
            self.torch_stream = stream if stream is not None else torch.cuda.current_stream(self.device)
            self._check(self.lib.urcco_session_create(index, C.c_void_p(self.torch_stream.cuda_stream), C.byref(handle)))
        else:
            # only meaningful when the binding points at the test-only host-simulator build
            self._check(self.lib.urcco_session_create(0, None, C.byref(handle)))
        self.handle = handle

    def _check(self, status: int):
        _lib.check(status, self.lib)

    def close(self):
        if self.handle:
            self.lib.urcco_session_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        self._check(self.lib.urcco_session_synchronize(self.handle))

    def set_timing(self, enable: bool):
        self._check(self.lib.urcco_session_set_timing(self.handle, int(enable)))

    def set_debug(self, flags: int):
        self._check(self.lib.urcco_session_set_debug(self.handle, int(flags)))

    def get_timings(self):
        """{stage name: (summed ms, launches)} since set_timing(True); synchronises."""
        ms = (C.c_double * _lib.N_STAGES)()
        n = (C.c_int64 * _lib.N_STAGES)()
        self._check(self.lib.urcco_session_get_timings(self.handle, ms, n))
        return {_lib.STAGE_NAMES[i]: (ms[i], n[i]) for i in range(_lib.N_STAGES) if _lib.STAGE_NAMES[i]}

    def empty(self, n, dtype):
        return torch.empty(int(n), dtype=dtype, device=self.device)

    # ---- stages ------------------------------------------------------------------------------------
    def column_counts(self, col_idx: torch.Tensor, nnz: int, n_cols: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        out = out if out is not None else self.empty(n_cols, torch.int32)
        self._check(self.lib.urcco_dev_column_counts(self.handle, nnz, _ptr(col_idx), n_cols, _ptr(out)))
        return out

    def downsample(self, m: DevCsr, nnz: int, raw_counts: torch.Tensor, seed: int, max_elements_per_row: int,
                   row_rate_mode: int = _lib.ROW_RATE_MAHOUT_INT_DIV, row_base: int = 0, post_out: Optional[torch.Tensor] = None):
        out_rp = self.empty(m.n_rows + 1, torch.int64)
        out_ci = self.empty(max(nnz, 1), torch.int32)
        post = post_out if post_out is not None else self.empty(max(m.n_cols, 1), torch.int32)
        self._check(self.lib.urcco_dev_downsample(self.handle, m.n_rows, _ptr(m.row_ptr), _ptr(m.col_idx), nnz, m.n_cols, _ptr(raw_counts),
                                                 _to_i32(seed), max_elements_per_row, row_rate_mode, row_base, _ptr(out_rp), _ptr(out_ci),
                                                 _ptr(post)))
        return DevCsr(m.n_rows, m.n_cols, out_rp, out_ci, nnz), post

    def transpose(self, m: DevCsr, counts: torch.Tensor, col_lo: int = 0, col_hi: Optional[int] = None):
        """CSC of m; only columns in [col_lo, col_hi) are materialised (the others are empty)."""
        col_hi = m.n_cols if col_hi is None else col_hi
        col_ptr = self.empty(m.n_cols + 1, torch.int64)
        row_idx = self.empty(max(m.nnz_bound, 1), torch.int32)
        self._check(self.lib.urcco_dev_transpose(self.handle, m.n_rows, _ptr(m.row_ptr), _ptr(m.col_idx), m.nnz_bound, m.n_cols, _ptr(counts),
                                                col_lo, col_hi, _ptr(col_ptr), _ptr(row_idx)))
        return col_ptr, row_idx

    def row_work_csr(self, a: DevCsr, b_row_ptr: torch.Tensor) -> torch.Tensor:
        """Per-item work contributed by this user shard (sum over ranks = row_work)."""
        work = self.empty(max(a.n_cols, 1), torch.int64)
        self._check(self.lib.urcco_dev_row_work_csr(self.handle, a.n_rows, _ptr(a.row_ptr), _ptr(a.col_idx), a.nnz_bound, _ptr(b_row_ptr), a.n_cols,
                                                   _ptr(work)))
        return work[: a.n_cols]

    def row_work(self, item_lo: int, item_hi: int, n_items_a: int, a_col_ptr, a_row_idx, nnz_a_bound: int, b_row_ptr) -> torch.Tensor:
        work = self.empty(max(item_hi - item_lo, 1), torch.int64)
        self._check(self.lib.urcco_dev_row_work(self.handle, item_lo, item_hi, n_items_a, _ptr(a_col_ptr), _ptr(a_row_idx), nnz_a_bound,
                                                _ptr(b_row_ptr), _ptr(work)))
        return work[: item_hi - item_lo]

    def partition(self, work: torch.Tensor, n_parts: int) -> List[int]:
        bounds = (C.c_int32 * (n_parts + 1))()
        self._check(self.lib.urcco_dev_partition(self.handle, work.numel(), _ptr(work), n_parts, bounds))
        return list(bounds)

    def cco_rows(self, item_lo: int, item_hi: int, n_items_a: int, a_col_ptr, a_row_idx, nnz_a_bound: int, b: DevCsr, counts_a, counts_b,
                 n_users: int, exclude_self: bool, p: DatasetParams) -> DevIndicators:
        n = item_hi - item_lo
        k = p.max_interesting_elements
        o_count = self.empty(max(n, 1), torch.int32)
        o_idx = self.empty(max(n * k, 1), torch.int32)
        o_llr = self.empty(max(n * k, 1), torch.float64)
        stats = self.empty(_lib.STATS_LEN, torch.int64)
        self._check(self.lib.urcco_dev_cco_rows(self.handle, item_lo, item_hi, n_items_a, _ptr(a_col_ptr), _ptr(a_row_idx), nnz_a_bound, _ptr(b.row_ptr),
                                               _ptr(b.col_idx), b.n_cols, _ptr(counts_a), _ptr(counts_b), n_users, int(exclude_self), k,
                                               int(p.min_llr is not None), float(p.min_llr) if p.min_llr is not None else 0.0,
                                               _ptr(o_count), _ptr(o_idx), _ptr(o_llr), _ptr(stats)))
        c_rp = self.empty(n + 1, torch.int64)
        c_idx = self.empty(max(n * k, 1), torch.int32)
        c_llr = self.empty(max(n * k, 1), torch.float64)
        self._check(self.lib.urcco_dev_compact_indicators(self.handle, n, k, _ptr(o_count), _ptr(o_idx), _ptr(o_llr), _ptr(c_rp), _ptr(c_idx),
                                                         _ptr(c_llr)))
        return DevIndicators(item_lo, item_hi, b.n_cols, k, c_rp, c_idx, c_llr, stats, b.row_ptr, b.col_idx)

    def llr(self, with_a, with_b, with_ab, n_users) -> torch.Tensor:
        out = self.empty(with_a.numel(), torch.float64)
        self._check(self.lib.urcco_dev_llr(self.handle, with_a.numel(), _ptr(with_a), _ptr(with_b), _ptr(with_ab), _ptr(n_users), _ptr(out)))
        return out

    def u01(self, seed: int, row, col) -> torch.Tensor:
        out = self.empty(row.numel(), torch.float64)
        self._check(self.lib.urcco_dev_u01(self.handle, row.numel(), _to_i32(seed), _ptr(row), _ptr(col), _ptr(out)))
        return out


def _to_i32(seed: int) -> int:
    """Scala `Long.toInt` (URAlgorithm.scala:240,325,345)."""
    s = int(seed) & 0xFFFFFFFF
    return s - (1 << 32) if s >= (1 << 31) else s


class SessionPool:
    """One urcco_session (own HIP stream + scratch arena) per event type, so that the per-event pipelines -- dozens of
    short kernels and persistent SpGEMM grids with ragged tails -- overlap on the GPU.  pool[0] runs the primary matrix."""

    def __init__(self, device: torch.device, n: int, library=None, priorities: Optional[Sequence[int]] = None):
        self.device = torch.device(device)
        if self.device.type == "cuda":
            pr = list(priorities) if priorities is not None else [0] * n
            self.sessions = [DeviceSession(self.device, library, torch.cuda.Stream(self.device, priority=pr[i])) for i in range(n)]
        else:
            self.sessions = [DeviceSession(self.device, library) for _ in range(n)]

    def __len__(self):
        return len(self.sessions)

    def __getitem__(self, i) -> DeviceSession:
        return self.sessions[i % len(self.sessions)]

    def close(self):
        for s in self.sessions:
            s.close()

    def synchronize(self):
        for s in self.sessions:
            s.synchronize()

    def set_timing(self, enable: bool):
        for s in self.sessions:
            s.set_timing(enable)

    def get_timings(self):
        out = {}
        for s in self.sessions:
            for k, (ms, n) in s.get_timings().items():
                a, b = out.get(k, (0.0, 0))
                out[k] = (a + ms, b + n)
        return out


def cross_occurrence_streams(pool: SessionPool, mats: Sequence[DevCsr], params: Sequence[DatasetParams], seed: int,
                             row_rate_mode: int = _lib.ROW_RATE_MAHOUT_INT_DIV) -> List[DevIndicators]:
    """cross_occurrence_device with one HIP stream per event type: B_i is sampled on its own stream while A is sampled and
    transposed on stream 0; every A'B_i then runs on stream i behind an event on A's CSC.  Same results, bit for bit."""
    if len(mats) == 0 or len(mats) != len(params):
        raise ValueError("need one DatasetParams per matrix and at least the primary matrix")
    a_raw = mats[0]
    for m in mats:
        if m.n_rows != a_raw.n_rows:
            raise ValueError("all matrices share the user dictionary: row counts differ")
    n_items_a = a_raw.n_cols
    if pool.device.type != "cuda":   # host-simulator sessions have no streams: plain sequential pipeline
        return cross_occurrence_device(pool[0], mats, params, seed, row_rate_mode)
    main = torch.cuda.current_stream(pool.device)
    streams = [pool[d].torch_stream for d in range(len(mats))]
    for st in set(streams):
        st.wait_stream(main)                       # inputs were produced on the caller's stream
    sampled = [None] * len(mats)
    with torch.cuda.stream(streams[0]):
        s0 = pool[0]
        raw = s0.column_counts(a_raw.col_idx, a_raw.nnz_bound, a_raw.n_cols)
        a, cnt_a = s0.downsample(a_raw, a_raw.nnz_bound, raw, seed, params[0].max_elements_per_row, row_rate_mode)
        a_col_ptr, a_row_idx = s0.transpose(a, cnt_a)
        a_ready = torch.cuda.Event()
        a_ready.record(streams[0])
        sampled[0] = (a, cnt_a)
    for d in range(1, len(mats)):
        with torch.cuda.stream(streams[d]):
            sd, m, p = pool[d], mats[d], params[d]
            raw_b = sd.column_counts(m.col_idx, m.nnz_bound, m.n_cols)
            sampled[d] = sd.downsample(m, m.nnz_bound, raw_b, seed, p.max_elements_per_row, row_rate_mode)
    out = [None] * len(mats)
    order = sorted(range(len(mats)), key=lambda d: -mats[d].nnz_bound)   # the heaviest event type is enqueued first
    for d in order:
        with torch.cuda.stream(streams[d]):
            if streams[d] is not streams[0]:
                streams[d].wait_event(a_ready)
                for t in (a_col_ptr, a_row_idx, cnt_a, a.row_ptr, a.col_idx):
                    t.record_stream(streams[d])   # produced on stream 0, read here
            b, cnt_b = sampled[d]
            ind = pool[d].cco_rows(0, n_items_a, n_items_a, a_col_ptr, a_row_idx, a.nnz_bound, b, cnt_a, cnt_b, a_raw.n_rows, d == 0, params[d])
            for t in (ind.row_ptr, ind.col_idx, ind.llr, ind.stats, b.row_ptr, b.col_idx):
                t.record_stream(main)             # consumed by the caller on its stream
            out[d] = ind
    for st in set(streams):
        main.wait_stream(st)
    return out


def cross_occurrence_device(sess: DeviceSession, mats: Sequence[DevCsr], params: Sequence[DatasetParams], seed: int,
                            row_rate_mode: int = _lib.ROW_RATE_MAHOUT_INT_DIV, item_lo: int = 0,
                            item_hi: Optional[int] = None) -> List[DevIndicators]:
    """SimilarityAnalysis.crossOccurrenceDownsampled on one GPU, inputs and outputs in HBM, no host sync.
    mats[0] is the primary matrix A; returns the indicator matrices for A'A, A'B_1, ..."""
    if len(mats) == 0 or len(mats) != len(params):
        raise ValueError("need one DatasetParams per matrix and at least the primary matrix")
    a_raw = mats[0]
    for m in mats:
        if m.n_rows != a_raw.n_rows:
            raise ValueError("all matrices share the user dictionary: row counts differ")
    if item_hi is None:
        item_hi = a_raw.n_cols
    raw = sess.column_counts(a_raw.col_idx, a_raw.nnz_bound, a_raw.n_cols)
    a, cnt_a = sess.downsample(a_raw, a_raw.nnz_bound, raw, seed, params[0].max_elements_per_row, row_rate_mode)
    a_col_ptr, a_row_idx = sess.transpose(a, cnt_a)
    out = []
    for d, (m, p) in enumerate(zip(mats, params)):
        if d == 0:
            b, cnt_b = a, cnt_a
        else:
            raw_b = sess.column_counts(m.col_idx, m.nnz_bound, m.n_cols)
            b, cnt_b = sess.downsample(m, m.nnz_bound, raw_b, seed, p.max_elements_per_row, row_rate_mode)
        out.append(sess.cco_rows(item_lo, item_hi, a_raw.n_cols, a_col_ptr, a_row_idx, a.nnz_bound, b, cnt_a, cnt_b, a_raw.n_rows, d == 0, p))
    return out
