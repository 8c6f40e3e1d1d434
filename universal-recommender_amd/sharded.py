"""Multi-GPU CCO model build: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The path shards with ONE exchange step:

  input phase   -- users are range-sharded: rank r holds rows [row_base, row_base + n_local) of every raw matrix.  Local
                   column counts of ALL event types go through one all-reduce (-> the raw counts sampleDownAndBinarize
                   needs), every shard is down-sampled (the RNG is keyed by the GLOBAL row, so the result does not
                   depend on the sharding), and the post-sampling counts go through a second all-reduce;
  ranges        -- items of A are split into world_size contiguous ranges of equal summed row work (not equal count:
                   Zipf skew).  The per-item work, summed over event types, is added up from the user shards by a third
                   all-reduce, so the ranges are known before any whole-matrix work; one range set serves every event
                   type (a rank's time is the sum over event types), hence ONE transposition of its slice of A';
  exchange      -- all-gathers of the down-sampled CSR shards (row lengths + column indices, padded to the largest
                   shard).  They are issued asynchronously in the order they are consumed -- A, then B_1, B_2, ... -- and
                   each is waited for only when its event type is about to be processed, so the gather of B_{d+1} runs
                   under the SpGEMM of B_d;
  compute phase -- each rank transposes and expands ONLY its item range of A' and emits the indicator rows of that
                   range -- disjoint rows, no further traffic.

Collectives per model build: 3 all-reduces (int32 counts x2, int64 work), 1 tiny all-gather of shard sizes, and per
event type 1 all-gather of row lengths + 1 of column indices.  There is one host synchronisation (shard sizes and range
bounds are read together).  Mahout does the same job with Spark broadcasts of the count vectors and a shuffle inside
`A.t %*% B` (reference call sites URAlgorithm.scala:323-346).  With world_size == 1 nothing is exchanged.
"""
from __future__ import annotations

import contextlib
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from . import _lib
from .device import DatasetParams, DevCsr, DevIndicators, DeviceSession


@dataclass
class ShardedResult:
    indicators: List[DevIndicators]       # this rank's rows, one entry per event type
    item_ranges: List[List[int]]          # per event type: world_size + 1 bounds
    nnz_sampled: List[int]                # global nnz after down-sampling, per event type


def _all_reduce_sum(t: torch.Tensor, group) -> None:
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def _exchange_sizes_start(locals_: Sequence[DevCsr], group) -> torch.Tensor:
    """(rows, nnz) of every rank's down-sampled shard for ALL event types in one tiny all-gather (device tensor: the
    caller reads it together with the range bounds, one host sync for both)."""
    world = dist.get_world_size(group)
    dev = locals_[0].row_ptr.device
    mine = torch.stack([v for m in locals_ for v in (torch.tensor(m.n_rows, dtype=torch.int64, device=dev), m.row_ptr[-1])])
    sizes = torch.empty(world * mine.numel(), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, mine, group=group)
    return sizes


def _exchange_sizes_finish(sizes: torch.Tensor, n_ds: int, group) -> List[List[List[int]]]:
    world = dist.get_world_size(group)
    sizes = sizes.cpu().view(world, n_ds, 2)
    return [[[int(sizes[r, d, 0]), int(sizes[r, d, 1])] for r in range(world)] for d in range(n_ds)]


@dataclass
class _PendingGather:
    local: DevCsr
    rows: List[int]
    nnzs: List[int]
    n_rows_global: int
    bufs: tuple            # (deg, all_deg, ci, all_ci): kept alive until the collectives have run
    works: tuple


def _gather_start(local: DevCsr, sizes: List[List[int]], n_rows_global: int, group) -> _PendingGather:
    """Issue the all-gather of a down-sampled row shard (row lengths as int32 + column indices, padded to the largest
    shard) without waiting for it.  sizes[r] = (rows, nnz) of rank r."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = local.row_ptr.device
    rows = [s[0] for s in sizes]
    nnzs = [s[1] for s in sizes]
    if sum(rows) != n_rows_global:
        raise ValueError(f"row shards sum to {sum(rows)} rows, expected {n_rows_global}")
    max_rows, max_nnz = max(max(rows), 1), max(max(nnzs), 1)
    if local.n_rows == max_rows:
        deg = torch.diff(local.row_ptr).to(torch.int32)
    else:
        deg = torch.zeros(max_rows, dtype=torch.int32, device=dev)
        deg[: local.n_rows] = torch.diff(local.row_ptr).to(torch.int32)
    all_deg = torch.empty(world * max_rows, dtype=torch.int32, device=dev)
    w1 = dist.all_gather_into_tensor(all_deg, deg, group=group, async_op=True)
    if local.col_idx.numel() >= max_nnz:
        ci = local.col_idx[:max_nnz]              # the shard's buffer is long enough: entries behind its nnz are never read
    else:
        ci = torch.zeros(max_nnz, dtype=torch.int32, device=dev)
        ci[: nnzs[rank]] = local.col_idx[: nnzs[rank]]
    all_ci = torch.empty(world * max_nnz, dtype=torch.int32, device=dev)
    w2 = dist.all_gather_into_tensor(all_ci, ci, group=group, async_op=True)
    return _PendingGather(local, rows, nnzs, n_rows_global, (deg, all_deg, ci, all_ci), (w1, w2))


def _gather_finish(p: _PendingGather) -> DevCsr:
    """Wait for the gather (the CURRENT stream waits, not the host) and assemble the whole matrix, rows in rank order."""
    for w in p.works:
        w.wait()
    _, all_deg, _, all_ci = p.bufs
    world = len(p.rows)
    dev = all_deg.device
    max_rows, max_nnz = all_deg.numel() // world, all_ci.numel() // world
    if world == 1:
        deg_cat, col_idx = all_deg[: p.rows[0]], all_ci[: max(p.nnzs[0], 1)]
    else:
        deg_cat = torch.cat([all_deg[r * max_rows: r * max_rows + p.rows[r]] for r in range(world)])
        col_idx = torch.cat([all_ci[r * max_nnz: r * max_nnz + p.nnzs[r]] for r in range(world)])
    row_ptr = torch.zeros(p.n_rows_global + 1, dtype=torch.int64, device=dev)
    torch.cumsum(deg_cat, 0, out=row_ptr[1:])
    total = sum(p.nnzs)
    if total == 0:
        col_idx = torch.zeros(1, dtype=torch.int32, device=dev)
    return DevCsr(p.n_rows_global, p.local.n_cols, row_ptr, col_idx, total)


def cross_occurrence_sharded(sess: DeviceSession, shards: Sequence[DevCsr], params: Sequence[DatasetParams], seed: int,
                             n_rows_global: int, row_base: int, row_rate_mode: int = _lib.ROW_RATE_MAHOUT_INT_DIV,
                             group=None, force_exchange: bool = False, pool=None) -> ShardedResult:
    """SimilarityAnalysis.crossOccurrenceDownsampled over world_size GPUs.  shards[d] = this rank's user rows of
    event type d (shards[0] = primary).  force_exchange runs the collectives and the range logic even in a one-rank
    group (used to exercise the RCCL path on a single GPU).  pool (device.SessionPool, optional): the A'B_d of each
    event type runs on its own HIP stream (as the single-GPU driver does), each behind its own gather; same results."""
    n_ranks = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    exchange = n_ranks > 1 or (force_exchange and dist.is_initialized())
    if len(shards) == 0 or len(shards) != len(params):
        raise ValueError("need one DatasetParams per matrix and at least the primary matrix")
    n_items_a = shards[0].n_cols
    n_ds = len(shards)

    dev = shards[0].row_ptr.device
    col_off = [0]
    for m in shards:
        col_off.append(col_off[-1] + max(m.n_cols, 1))

    # ---- input phase: raw counts of every event type -> one all-reduce -> down-sampling -> one all-reduce of the
    #      post-sampling counts (nothing here waits for the host).  With a pool the per-event-type work runs on the
    #      event type's own HIP stream; the collectives join them on the caller's stream.
    use_streams = pool is not None and dev.type == "cuda"
    main = torch.cuda.current_stream(dev) if use_streams else None
    streams = [pool[d].torch_stream for d in range(n_ds)] if use_streams else []

    def fork():
        for st in set(streams):
            st.wait_stream(main)

    def join():
        for st in set(streams):
            main.wait_stream(st)

    def on(d):
        return torch.cuda.stream(streams[d]) if use_streams else contextlib.nullcontext()

    def worker(d):
        return pool[d] if use_streams else sess

    raw_all = torch.empty(col_off[-1], dtype=torch.int32, device=dev)
    post_all = torch.empty(col_off[-1], dtype=torch.int32, device=dev)
    for st in set(streams):
        raw_all.record_stream(st)
        post_all.record_stream(st)
    fork()
    for d, m in enumerate(shards):
        with on(d):
            worker(d).column_counts(m.col_idx, m.nnz_bound, m.n_cols, out=raw_all[col_off[d]: col_off[d + 1]])
    join()
    if exchange:
        _all_reduce_sum(raw_all, group)
    locals_: List[DevCsr] = []
    counts: List[torch.Tensor] = []
    fork()
    for d, (m, p) in enumerate(zip(shards, params)):
        with on(d):
            local, post = worker(d).downsample(m, m.nnz_bound, raw_all[col_off[d]: col_off[d + 1]], seed, p.max_elements_per_row, row_rate_mode,
                                               row_base, post_out=post_all[col_off[d]: col_off[d + 1]])
            if use_streams:
                local.row_ptr.record_stream(main)     # allocated on stream d, read by the collectives / compute phase on main
                local.col_idx.record_stream(main)
        locals_.append(local)
        counts.append(post)
    join()
    if exchange:
        _all_reduce_sum(post_all, group)

    if not exchange:  # (kept for callers that pass a plain session; bench.py uses device.cross_occurrence_streams at N = 1)
        a = locals_[0]
        a_col_ptr, a_row_idx = sess.transpose(a, counts[0])
        out = [sess.cco_rows(0, n_items_a, n_items_a, a_col_ptr, a_row_idx, a.nnz_bound, locals_[d], counts[0], counts[d], n_rows_global, d == 0,
                             params[d]) for d in range(n_ds)]
        return ShardedResult(out, [[0, n_items_a]] * n_ds, [-1] * n_ds)

    # ---- work-balanced item ranges, fixed BEFORE any whole-matrix work: every rank adds up the row work its own users
    #      contribute (summed over event types), one all-reduce makes it global, the same prefix split runs on every
    #      rank.  The shard sizes travel at the same time; both are read by the one host sync of the build.
    #      (Row work is linear in the B row lengths, so the sum over event types is ONE pass over the A shard against
    #      the element-wise sum of the B row_ptr arrays -- one atomic per interaction of A instead of one per event type.)
    rp_sum = locals_[0].row_ptr if n_ds == 1 else torch.stack([m.row_ptr for m in locals_]).sum(0)
    work = sess.row_work_csr(locals_[0], rp_sum)
    _all_reduce_sum(work, group)
    sizes_dev = _exchange_sizes_start(locals_, group)
    bounds = sess.partition(work, n_ranks)                    # synchronises the stream
    sizes = _exchange_sizes_finish(sizes_dev, n_ds, group)
    # ---- exchange: all gathers are issued now, each is waited for (by the stream) when its event type comes up
    pending = [_gather_start(locals_[d], sizes[d], n_rows_global, group) for d in range(n_ds)]
    # ---- compute phase: a rank transposes and expands only the item range it owns
    lo, hi = bounds[rank], bounds[rank + 1]
    a = _gather_finish(pending[0])
    a_col_ptr, a_row_idx = sess.transpose(a, counts[0], lo, hi)
    out: List[Optional[DevIndicators]] = [None] * n_ds
    nnz_sampled = [sum(sz[1] for sz in sizes[d]) for d in range(n_ds)]
    if not use_streams:
        for d in range(n_ds):
            b = a if d == 0 else _gather_finish(pending[d])
            out[d] = sess.cco_rows(lo, hi, n_items_a, a_col_ptr, a_row_idx, a.nnz_bound, b, counts[0], counts[d], n_rows_global, d == 0, params[d])
        return ShardedResult(out, [list(bounds)] * n_ds, nnz_sampled)
    # one HIP stream per event type: stream d waits for A's CSC slice and for its own gather only
    a_ready = torch.cuda.Event()
    a_ready.record(main)
    for d in sorted(range(n_ds), key=lambda d: -nnz_sampled[d]):   # the heaviest event type is enqueued first
        st = streams[d]
        with torch.cuda.stream(st):
            st.wait_event(a_ready)
            for t in (a_col_ptr, a_row_idx, counts[0], counts[d], a.row_ptr, a.col_idx) + (pending[d].bufs if d else ()):
                t.record_stream(st)               # allocated on the caller's stream, read here
            b = a if d == 0 else _gather_finish(pending[d])
            ind = pool[d].cco_rows(lo, hi, n_items_a, a_col_ptr, a_row_idx, a.nnz_bound, b, counts[0], counts[d], n_rows_global, d == 0, params[d])
            for t in (ind.row_ptr, ind.col_idx, ind.llr, ind.stats, b.row_ptr, b.col_idx):
                t.record_stream(main)             # consumed by the caller on its stream
            out[d] = ind
    for st in set(streams):
        main.wait_stream(st)
    return ShardedResult(out, [list(bounds)] * n_ds, nnz_sampled)


def gather_indicators_to_host(res: ShardedResult, group=None):
    """Concatenate every rank's indicator rows on every rank (host numpy): list of (row_ptr, col_idx, llr).
    Not part of the timed model build (the reference hands the rows to URModel.save per partition)."""
    import numpy as np
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    out = []
    for ind in res.indicators:
        rp, ci, llr = ind.to_host()
        if world == 1:
            out.append((rp, ci, llr))
            continue
        parts = [None] * world
        dist.all_gather_object(parts, (np.diff(rp), ci, llr), group=group)
        lens = np.concatenate([p[0] for p in parts])
        full_rp = np.zeros(lens.size + 1, np.int64)
        np.cumsum(lens, out=full_rp[1:])
        out.append((full_rp, np.concatenate([p[1] for p in parts]), np.concatenate([p[2] for p in parts])))
    return out
