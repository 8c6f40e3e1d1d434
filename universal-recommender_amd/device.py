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
    sampled_row_ptr: Optional[torch.Tensor] = None  # row_ptr of the down-sampled B this GPU multiplied with (several ranks: only the rows its
                                                    # item range touches are filled, include/urcco.h urcco_dev_result)
    sampled_col_idx: Optional[torch.Tensor] = None  # col_idx of the same (first sampled_row_ptr[-1] entries live)
    sampled_nnz_total: int = -1                     # entries of the WHOLE down-sampled matrix over all ranks (-1: one rank, = sampled_row_ptr[-1])

    def nnz_sampled_global(self) -> int:
        """nnz' of the whole matrix: the library's host-side total when the build exchanged, else the (whole) matrix this GPU holds."""
        if self.sampled_nnz_total >= 0:
            return int(self.sampled_nnz_total)
        return int(self.sampled_row_ptr[-1]) if self.sampled_row_ptr is not None and self.sampled_row_ptr.numel() else 0

    def to_host(self):
        if self.row_ptr.numel() == 0:
            import numpy as np
            return np.zeros(1, np.int64), np.zeros(0, np.int32), np.zeros(0, np.float64)
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
        self.pack_counts = True   # cco_rows: B' with the columns' counts aboard (False: one count gather per candidate, the form of rounds 1-5)
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

    def merge_fragments(self, world: int, item_lo: int, item_hi: int, n_items: int, lens: torch.Tensor, entries: torch.Tensor, n_entries: int,
                        sizes: torch.Tensor, counts: torch.Tensor):
        """CSC of the item range [item_lo, item_hi) from the fragments of `world` user shards (urcco_dev_merge_fragments):
        lens = uint16 or int32 column lengths, rank-major; entries = shard-local user ids, the fragments in rank order."""
        assert lens.dtype in (torch.uint16, torch.int16, torch.int32)
        col_ptr = self.empty(n_items + 1, torch.int64)
        row_idx = self.empty(max(n_entries, 1), torch.int32)
        self._check(self.lib.urcco_dev_merge_fragments(self.handle, world, item_lo, item_hi, n_items, _ptr(lens), int(lens.dtype != torch.int32), _ptr(entries),
                                                      n_entries, _ptr(sizes), _ptr(counts), _ptr(col_ptr), _ptr(row_idx)))
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
        # B' with the columns' counts aboard (round 6; what the context level does for every build): the row kernels read a candidate's cB off the
        # word that claims its accumulator slot.  pack_counts=False: the plain entry point with one count gather per candidate (rounds 1-5)
        if self.pack_counts and b.n_cols > 0:
            packed = self.empty(max(b.nnz_bound, 1), torch.int32)
            bad = self.empty(1, torch.int32)
            self._check(self.lib.urcco_dev_pack_counts(self.handle, b.n_rows, _ptr(b.row_ptr), _ptr(b.col_idx), b.nnz_bound, _ptr(counts_b), b.n_cols,
                                                      _ptr(packed), _ptr(bad)))
            self._check(self.lib.urcco_dev_cco_rows_packed(self.handle, item_lo, item_hi, n_items_a, _ptr(a_col_ptr), _ptr(a_row_idx), nnz_a_bound, _ptr(b.row_ptr),
                                                          _ptr(b.col_idx), b.n_cols, _ptr(counts_a), _ptr(counts_b), n_users, int(exclude_self), k,
                                                          int(p.min_llr is not None), float(p.min_llr) if p.min_llr is not None else 0.0,
                                                          _ptr(o_count), _ptr(o_idx), _ptr(o_llr), _ptr(stats), _ptr(packed), _ptr(bad)))
        else:
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

    def u01(self, seed: int, row, col, rng: int = _lib.RNG_SPLITMIX53) -> torch.Tensor:
        out = self.empty(row.numel(), torch.float64)
        if rng == _lib.RNG_SPLITMIX53:
            self._check(self.lib.urcco_dev_u01(self.handle, row.numel(), _to_i32(seed), _ptr(row), _ptr(col), _ptr(out)))
        else:
            self._check(self.lib.urcco_dev_u01_rng(self.handle, row.numel(), _to_i32(seed), _ptr(row), _ptr(col), int(rng), _ptr(out)))
        return out


def _to_i32(seed: int) -> int:
    """Scala `Long.toInt` (URAlgorithm.scala:240,325,345)."""
    s = int(seed) & 0xFFFFFFFF
    return s - (1 << 32) if s >= (1 << 31) else s


_TYPESTR = {torch.int32: "<i4", torch.int64: "<i8", torch.float64: "<f8", torch.uint8: "|u1"}
_CTYPE = {torch.int32: C.c_int32, torch.int64: C.c_int64, torch.float64: C.c_double, torch.uint8: C.c_uint8}


class _RawDeviceArray:
    """Zero-copy handle on context-owned device memory (CUDA array interface, which PyTorch-ROCm honours)."""

    def __init__(self, ptr: int, n: int, dtype: torch.dtype):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": _TYPESTR[dtype], "data": (ptr, False), "version": 2}


def _view(ptr: Optional[int], n: int, dtype: torch.dtype, device: torch.device) -> torch.Tensor:
    """Tensor aliasing n elements at a raw pointer of the context (valid until its next build)."""
    if not ptr or n <= 0:
        return torch.empty(0, dtype=dtype, device=device)
    if device.type == "cuda":
        return torch.as_tensor(_RawDeviceArray(ptr, n, dtype), device=device)
    import numpy as np
    return torch.from_numpy(np.ctypeslib.as_array((_CTYPE[dtype] * n).from_address(ptr)))


class Context:
    """urcco_context (include/urcco.h, CONTEXT level): the persistent, multi-stream, multi-GPU form of the model build.

    One context = this process's GPUs [device, device + n_gpus) as ranks [first_rank, first_rank + n_gpus) of a job of
    world_size ranks.  Collectives are RCCL inside the library (nccl_unique_id from `unique_id()` on one rank when the
    ranks are spread over processes) unless `collectives` replaces them (the CPU test-suite: gloo on the simulator)."""

    def __init__(self, device, library=None, n_gpus: int = 1, flags: int = 0, row_rate_mode: int = _lib.ROW_RATE_MAHOUT_INT_DIV,
                 world_size: int = 0, first_rank: int = 0, unique_id: Optional[bytes] = None, collectives=None):
        self.device = torch.device(device)
        self.lib = library if library is not None else _lib.lib()
        index = 0
        if self.device.type == "cuda":
            index = self.device.index if self.device.index is not None else torch.cuda.current_device()
            self.device = torch.device("cuda", index)
        emulate = bool(flags & _lib.FLAG_EMULATE_RANKS)  # measurement only: every rank on the ONE device (include/urcco.h)
        self.devices = [torch.device("cuda", index + (0 if emulate else g)) if self.device.type == "cuda" else self.device for g in range(max(n_gpus, 1))]
        opts = _lib.Options(device=index, row_rate_mode=row_rate_mode, n_gpus=n_gpus, flags=flags)
        comm = None
        self._collectives = collectives
        self._keep = [collectives]
        if world_size or first_rank or unique_id is not None or collectives is not None:
            comm = _lib.CommConfig(world_size=world_size, first_rank=first_rank)
            if unique_id is not None:
                buf = C.create_string_buffer(bytes(unique_id), _lib.UNIQUE_ID_BYTES)
                self._keep.append(buf)
                comm.nccl_unique_id = C.cast(buf, C.c_void_p)
            if collectives is not None:
                comm.collectives = C.pointer(collectives.struct)
        handle = C.c_void_p()
        _lib.check(self.lib.urcco_context_create(C.byref(opts), C.byref(comm) if comm is not None else None, C.byref(handle)), self.lib)
        self.handle = handle
        self.n_local = int(self.lib.urcco_context_local_gpus(handle))
        self.devices = self.devices[: self.n_local] if len(self.devices) >= self.n_local else [torch.device("cuda", index + g) for g in range(self.n_local)]
        self._last = None

    @staticmethod
    def unique_id(library=None) -> bytes:
        lib = library if library is not None else _lib.lib()
        buf = C.create_string_buffer(_lib.UNIQUE_ID_BYTES)
        _lib.check(lib.urcco_comm_unique_id(buf), lib)
        return buf.raw

    def collectives_error(self):
        return getattr(self._collectives, "error", None)

    def _check(self, status: int):
        if status != _lib.OK and self.collectives_error() is not None:
            raise self.collectives_error()      # the Python exception behind a failed collectives callback
        _lib.check(status, self.lib)

    def close(self):
        if self.handle:
            self.lib.urcco_context_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        self._check(self.lib.urcco_context_synchronize(self.handle))

    def set_timing(self, enable: bool):
        self._check(self.lib.urcco_context_set_timing(self.handle, int(enable)))

    def set_debug(self, flags: int):
        self._check(self.lib.urcco_context_set_debug(self.handle, int(flags)))

    def set_flags(self, flags: int):
        self._check(self.lib.urcco_context_set_flags(self.handle, int(flags)))

    def get_timings(self):
        ms = (C.c_double * _lib.N_STAGES)()
        n = (C.c_int64 * _lib.N_STAGES)()
        self._check(self.lib.urcco_context_get_timings(self.handle, ms, n))
        return {_lib.STAGE_NAMES[i]: (ms[i], n[i]) for i in range(_lib.N_STAGES) if _lib.STAGE_NAMES[i]}

    def get_timings_gpu(self, local_gpu: int):
        """The stage timings of ONE local GPU (rank first_rank + local_gpu)."""
        ms = (C.c_double * _lib.N_STAGES)()
        n = (C.c_int64 * _lib.N_STAGES)()
        self._check(self.lib.urcco_context_get_timings_gpu(self.handle, int(local_gpu), ms, n))
        return {_lib.STAGE_NAMES[i]: (ms[i], n[i]) for i in range(_lib.N_STAGES) if _lib.STAGE_NAMES[i]}

    def build(self, shards: Sequence[Sequence[DevCsr]], params: Sequence[DatasetParams], seed: int, n_users_total: Optional[int] = None,
              row_bases: Optional[Sequence[int]] = None, input_stream=None):
        """Enqueue one model build.  shards[d][g] = event type d's user rows held by local GPU g (shards[d] may also be a
        single DevCsr for a one-GPU context).  Returns immediately (after the one blocking read of ranges / sizes with
        more than one rank); read the outcome with results()."""
        n_ds = len(shards)
        if n_ds == 0 or n_ds != len(params):
            raise ValueError("need one DatasetParams per matrix and at least the primary matrix")
        per = [list(sd) if isinstance(sd, (list, tuple)) else [sd] for sd in shards]
        L = self.n_local
        if any(len(p) != L for p in per):
            raise ValueError(f"every event type needs one shard per local GPU ({L})")
        if row_bases is None:
            row_bases, acc = [], 0
            for g in range(L):
                row_bases.append(acc)
                acc += per[0][g].n_rows
        if n_users_total is None:
            n_users_total = sum(m.n_rows for m in per[0])
        ds = (_lib.DevDataset * n_ds)()
        keep = []
        for d in range(n_ds):
            arr = (_lib.DevShard * L)()
            for g, m in enumerate(per[d]):
                arr[g].n_rows, arr[g].row_base, arr[g].nnz = m.n_rows, row_bases[g], m.nnz_bound
                arr[g].row_ptr, arr[g].col_idx = m.row_ptr.data_ptr(), m.col_idx.data_ptr()
            keep.append(arr)
            p = params[d]
            ds[d].n_cols = per[d][0].n_cols
            ds[d].max_elements_per_row, ds[d].max_interesting_elements = p.max_elements_per_row, p.max_interesting_elements
            ds[d].has_min_llr, ds[d].min_llr = int(p.min_llr is not None), float(p.min_llr) if p.min_llr is not None else 0.0
            ds[d].shards = arr
        out = (_lib.DevResult * (n_ds * L))()
        st = None
        if input_stream is not None:
            st = C.c_void_p(input_stream.cuda_stream)
        elif self.device.type == "cuda" and L == 1:
            st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        self._check(self.lib.urcco_context_build_device(self.handle, ds, n_ds, int(n_users_total), _to_i32(seed), st, out))
        self._last = (out, n_ds, [p.max_interesting_elements for p in params], [per[d][0].n_cols for d in range(n_ds)], (per, keep))
        return out

    def wait(self):
        """torch's current stream waits (on the device) for the last build (single-GPU contexts)."""
        if self.device.type == "cuda" and self.n_local == 1:
            self._check(self.lib.urcco_context_wait_stream(self.handle, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        else:
            self.synchronize()

    def results(self) -> List[List[DevIndicators]]:
        """Views on the last build's outputs (context-owned memory, valid until the next build): [event type][local GPU].
        Synchronises."""
        self.synchronize()
        out, n_ds, ks, n_cols, _ = self._last
        L = self.n_local
        res = []
        for d in range(n_ds):
            row = []
            for g in range(L):
                r = out[d * L + g]
                dev = self.devices[g]
                n = r.item_hi - r.item_lo
                s_rp = _view(r.sampled_row_ptr, r.sampled_rows + 1, torch.int64, dev)
                s_nnz = int(s_rp[-1]) if s_rp.numel() else 0
                s_ci = _view(r.sampled_col_idx, max(s_nnz, 1), torch.int32, dev)
                if r.sampled_col_mask != -1:   # a sharded build's rows arrive with their columns' counts above the column bits (include/urcco.h)
                    s_ci = s_ci & int(r.sampled_col_mask)
                row.append(DevIndicators(r.item_lo, r.item_hi, n_cols[d], ks[d], _view(r.row_ptr, n + 1, torch.int64, dev),
                                         _view(r.col_idx, max(n * ks[d], 1), torch.int32, dev), _view(r.llr, max(n * ks[d], 1), torch.float64, dev),
                                         _view(r.stats, _lib.STATS_LEN, torch.int64, dev), s_rp, s_ci,
                                         int(r.sampled_nnz_total)))
            res.append(row)
        return res


def cross_occurrence_context(ctx: Context, mats: Sequence[DevCsr], params: Sequence[DatasetParams], seed: int) -> List[DevIndicators]:
    """SimilarityAnalysis.crossOccurrenceDownsampled on a one-GPU context: one build, results as views."""
    ctx.build([[m] for m in mats], params, seed)
    return [r[0] for r in ctx.results()]


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
