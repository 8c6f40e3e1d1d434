"""Multi-GPU CCO model build: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The path shards with ONE exchange step:

  input phase   -- users are range-sharded: rank r holds rows [row_base, row_base + n_local) of every raw matrix.  Per
                   event type: local column counts -> all-reduce (the raw counts sampleDownAndBinarize needs) ->
                   down-sampling of the shard (the RNG is keyed by the GLOBAL row, so the result does not depend on
                   the sharding) -> all-reduce of the post-sampling counts.  Each event type runs on its own HIP stream
                   when a SessionPool is given; collectives synchronise with that stream only;
  ranges        -- items of A are split into world_size contiguous ranges of equal summed row work (not equal count:
                   Zipf skew).  The key is the A'A row work, added up from the user shards by one all-reduce as soon as
                   A is sampled (the work of A'B_d sums the same users' B_d row lengths and follows it closely), so the
                   ranges are known before any whole-matrix work and the blocking host read of the build (range bounds
                   + A's shard sizes) comes after the primary's short chain only; the secondary event types are sampled
                   afterwards, under the SpGEMM of A'A.  One range set serves every event type, hence ONE
                   transposition of the rank's slice of A';
  exchange      -- all-gathers of the down-sampled CSR shards (row lengths + column indices, padded to the largest
                   shard), issued asynchronously per event type: A first, B_d on stream d behind its own sampling, so
                   the gathers of the secondaries run under the SpGEMM of A'A;
  compute phase -- each rank transposes and expands ONLY its item range of A' and emits the indicator rows of that
                   range -- disjoint rows, no further traffic.

Collectives per model build: per event type 2 all-reduces (int32 counts), 1 tiny all-gather of shard sizes, 1
all-gather of row lengths + 1 of column indices; plus 1 all-reduce of the int64 work key.  The host blocks once on the
primary's stream (bounds + sizes of A); the secondaries' shard sizes are read when their gathers are issued, by which time
the GPU is busy with A'A.  Mahout does the same job with Spark broadcasts of the count vectors and a shuffle inside
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
    use_streams = pool is not None and dev.type == "cuda"
    main = torch.cuda.current_stream(dev) if use_streams else None
    streams = [pool[d].torch_stream for d in range(n_ds)] if use_streams else []

    def on(d):
        """Event type d's stream (collectives issued inside synchronise with that stream only)."""
        return torch.cuda.stream(streams[d]) if use_streams else contextlib.nullcontext()

    def worker(d) -> DeviceSession:
        return pool[d] if use_streams else sess

    def to_main(*tensors):
        if use_streams:
            for t in tensors:
                t.record_stream(main)

    for st in set(streams):
        st.wait_stream(main)                          # the shards were produced on the caller's stream

    def input_phase(d):
        """raw counts -> all-reduce -> down-sampling -> all-reduce of the post-sampling counts; nothing waits for the host"""
        m, p = shards[d], params[d]
        w = worker(d)
        raw = w.column_counts(m.col_idx, m.nnz_bound, m.n_cols)
        if exchange:
            _all_reduce_sum(raw, group)
        local, post = w.downsample(m, m.nnz_bound, raw, seed, p.max_elements_per_row, row_rate_mode, row_base)
        if exchange:
            _all_reduce_sum(post, group)
        to_main(local.row_ptr, local.col_idx, post)
        return local, post

    locals_: List[Optional[DevCsr]] = [None] * n_ds
    counts: List[Optional[torch.Tensor]] = [None] * n_ds

    if not exchange:  # (kept for callers that pass a plain session; bench.py uses device.cross_occurrence_streams at N = 1)
        for d in range(n_ds):
            with on(d):
                locals_[d], counts[d] = input_phase(d)
        for st in set(streams):
            main.wait_stream(st)
        a = locals_[0]
        a_col_ptr, a_row_idx = sess.transpose(a, counts[0])
        out = [sess.cco_rows(0, n_items_a, n_items_a, a_col_ptr, a_row_idx, a.nnz_bound, locals_[d], counts[0], counts[d], n_rows_global, d == 0,
                             params[d]) for d in range(n_ds)]
        return ShardedResult(out, [[0, n_items_a]] * n_ds, [-1] * n_ds)

    # ---- primary event type first: its input phase, the balance key and the shard sizes of A
    with on(0):
        locals_[0], counts[0] = input_phase(0)
        # Work-balanced item ranges, fixed BEFORE any whole-matrix work and before the secondary event types are even
        # sampled: the key is the A'A row work, summed from the user shards by one all-reduce.  (The work of A'B_d for
        # item i is the sum over the same users of their B_d row lengths, so it follows the A'A key closely; using it
        # as the proxy lets the one blocking host read come right after the primary's short chain.)
        work = worker(0).row_work_csr(locals_[0], locals_[0].row_ptr)
        _all_reduce_sum(work, group)
        sizes_dev: List[Optional[torch.Tensor]] = [None] * n_ds
        sizes_dev[0] = _exchange_sizes_start([locals_[0]], group)
    # ---- ranges + exchange + compute of the primary.  (The secondaries are enqueued only afterwards: measured on one
    #      GPU, sampling them concurrently delays the primary's short dependent chain -- and with it the host read every
    #      rank blocks on -- by more than it saves; behind A'A they fill the SpGEMM kernels' ragged tails instead.)
    out: List[Optional[DevIndicators]] = [None] * n_ds
    nnz_sampled = [0] * n_ds
    with on(0):
        bounds = worker(0).partition(work, n_ranks)                  # synchronises stream 0 (the host read)
        lo, hi = bounds[rank], bounds[rank + 1]
        sizes0 = _exchange_sizes_finish(sizes_dev[0], 1, group)[0]
        nnz_sampled[0] = sum(sz[1] for sz in sizes0)
        a = _gather_finish(_gather_start(locals_[0], sizes0, n_rows_global, group))
        a_col_ptr, a_row_idx = worker(0).transpose(a, counts[0], lo, hi)
        if use_streams:
            a_ready = torch.cuda.Event()
            a_ready.record(streams[0])
        out[0] = worker(0).cco_rows(lo, hi, n_items_a, a_col_ptr, a_row_idx, a.nnz_bound, a, counts[0], counts[0], n_rows_global, True, params[0])
        to_main(out[0].row_ptr, out[0].col_idx, out[0].llr, out[0].stats, a.row_ptr, a.col_idx)
    # ---- secondaries: every input phase is enqueued on its own stream (they run under A'A); then, per event type, the
    #      shard sizes are read (the host waits for that stream's sampling only, the GPU stays busy), the gather is issued
    #      and A'B_d runs behind it and behind A's CSC slice
    for d in range(1, n_ds):
        with on(d):
            locals_[d], counts[d] = input_phase(d)
            sizes_dev[d] = _exchange_sizes_start([locals_[d]], group)
    for d in range(1, n_ds):
        with on(d):
            sizes_d = _exchange_sizes_finish(sizes_dev[d], 1, group)[0]
            nnz_sampled[d] = sum(sz[1] for sz in sizes_d)
            pending = _gather_start(locals_[d], sizes_d, n_rows_global, group)
            if use_streams:
                streams[d].wait_event(a_ready)
                for t in (a_col_ptr, a_row_idx, counts[0], a.row_ptr, a.col_idx):
                    t.record_stream(streams[d])               # produced on stream 0, read here
            b = _gather_finish(pending)
            out[d] = worker(d).cco_rows(lo, hi, n_items_a, a_col_ptr, a_row_idx, a.nnz_bound, b, counts[0], counts[d], n_rows_global, False, params[d])
            to_main(out[d].row_ptr, out[d].col_idx, out[d].llr, out[d].stats, b.row_ptr, b.col_idx)
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
