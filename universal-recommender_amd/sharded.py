"""Multi-GPU CCO model build: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The path shards with ONE exchange step:

  input phase   -- users are range-sharded: rank r holds rows [row_base, row_base + n_local) of every raw matrix,
                   computes local column counts (all-reduce -> the raw counts sampleDownAndBinarize needs), down-samples
                   its rows (the RNG is keyed by the GLOBAL row, so the result does not depend on the sharding) and
                   all-reduces the post-sampling column counts;
  exchange      -- all-gather of the down-sampled CSR shards (variable length: padded to the largest shard, then
                   compacted), after which every rank holds A' and each B'_i whole;
  compute phase -- items of A are split into world_size contiguous ranges of equal summed row work (not equal count:
                   Zipf skew; the per-item work is summed from the user shards by one all-reduce, so the ranges are known
                   before any whole-matrix work), each rank transposes and expands ONLY its item range of A' and emits
                   the indicator rows of that range -- disjoint rows, no further traffic.

Collectives per event type: 2 small all-reduces (int32[n_items]) + 1 all-reduce of the row work (int64[n_items]) +
1 all-gather of row lengths + 1 all-gather of column indices.  Mahout does the same job with Spark broadcasts of the count vectors and a shuffle inside `A.t %*% B`
(reference call sites URAlgorithm.scala:323-346).  With world_size == 1 nothing is exchanged.
"""
from __future__ import annotations

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


def _exchange_sizes(locals_: Sequence[DevCsr], group) -> List[List[List[int]]]:
    """(rows, nnz) of every rank's down-sampled shard for ALL event types in one tiny all-gather: the first of the two
    host syncs of a multi-GPU build (it sizes the receive buffers of the exchange)."""
    world = dist.get_world_size(group)
    dev = locals_[0].row_ptr.device
    mine = torch.stack([v for m in locals_ for v in (torch.tensor(m.n_rows, dtype=torch.int64, device=dev), m.row_ptr[-1])])
    sizes = torch.empty(world * mine.numel(), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, mine, group=group)
    sizes = sizes.cpu().view(world, len(locals_), 2)
    return [[[int(sizes[r, d, 0]), int(sizes[r, d, 1])] for r in range(world)] for d in range(len(locals_))]


def _gather_sampled(local: DevCsr, sizes: List[List[int]], n_rows_global: int, group) -> DevCsr:
    """All-gather a down-sampled row shard into the whole matrix (rows in rank order).  sizes[r] = (rows, nnz) of rank r."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = local.row_ptr.device
    rows = [s[0] for s in sizes]
    nnzs = [s[1] for s in sizes]
    if sum(rows) != n_rows_global:
        raise ValueError(f"row shards sum to {sum(rows)} rows, expected {n_rows_global}")
    max_rows, max_nnz = max(rows), max(max(nnzs), 1)
    # row lengths (int32) and column indices, padded to the largest shard
    deg = torch.zeros(max_rows, dtype=torch.int32, device=dev)
    deg[: local.n_rows] = (local.row_ptr[1:] - local.row_ptr[:-1]).to(torch.int32)
    all_deg = torch.empty(world * max_rows, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(all_deg, deg, group=group)
    ci = torch.zeros(max_nnz, dtype=torch.int32, device=dev)
    ci[: nnzs[rank]] = local.col_idx[: nnzs[rank]]
    all_ci = torch.empty(world * max_nnz, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(all_ci, ci, group=group)
    if world == 1:
        deg_cat, col_idx = all_deg[: rows[0]], all_ci[: max(nnzs[0], 1)]
    else:
        deg_cat = torch.cat([all_deg[r * max_rows: r * max_rows + rows[r]] for r in range(world)])
        col_idx = torch.cat([all_ci[r * max_nnz: r * max_nnz + nnzs[r]] for r in range(world)])
    row_ptr = torch.zeros(n_rows_global + 1, dtype=torch.int64, device=dev)
    torch.cumsum(deg_cat, 0, out=row_ptr[1:])
    total = sum(nnzs)
    if total == 0:
        col_idx = torch.zeros(1, dtype=torch.int32, device=dev)
    return DevCsr(n_rows_global, local.n_cols, row_ptr, col_idx, total)


def cross_occurrence_sharded(sess: DeviceSession, shards: Sequence[DevCsr], params: Sequence[DatasetParams], seed: int,
                             n_rows_global: int, row_base: int, row_rate_mode: int = _lib.ROW_RATE_MAHOUT_INT_DIV,
                             group=None, force_exchange: bool = False) -> ShardedResult:
    """SimilarityAnalysis.crossOccurrenceDownsampled over world_size GPUs.  shards[d] = this rank's user rows of
    event type d (shards[0] = primary).  force_exchange runs the collectives and the range logic even in a one-rank
    group (used to exercise the RCCL path on a single GPU)."""
    n_ranks = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    exchange = n_ranks > 1 or (force_exchange and dist.is_initialized())
    if len(shards) == 0 or len(shards) != len(params):
        raise ValueError("need one DatasetParams per matrix and at least the primary matrix")
    n_items_a = shards[0].n_cols
    n_ds = len(shards)

    # ---- input phase: every event type's shard is down-sampled (nothing here waits for the host)
    locals_: List[DevCsr] = []
    counts: List[torch.Tensor] = []
    for m, p in zip(shards, params):
        raw = sess.column_counts(m.col_idx, m.nnz_bound, m.n_cols)
        if exchange:
            _all_reduce_sum(raw, group)
        local, post = sess.downsample(m, m.nnz_bound, raw, seed, p.max_elements_per_row, row_rate_mode, row_base)
        if exchange:
            _all_reduce_sum(post, group)
        locals_.append(local)
        counts.append(post)

    if not exchange:  # (kept for callers that pass a plain session; bench.py uses device.cross_occurrence_streams at N = 1)
        a = locals_[0]
        a_col_ptr, a_row_idx = sess.transpose(a, counts[0])
        out = [sess.cco_rows(0, n_items_a, n_items_a, a_col_ptr, a_row_idx, a.nnz_bound, locals_[d], counts[0], counts[d], n_rows_global, d == 0,
                             params[d]) for d in range(n_ds)]
        return ShardedResult(out, [[0, n_items_a]] * n_ds, [-1] * n_ds)

    # ---- work-balanced item ranges, fixed BEFORE any whole-matrix work: every rank adds up the row work its own users
    #      contribute (per event type), one all-reduce makes it global, the same prefix split runs on every rank.
    works = []
    for d in range(n_ds):
        w = sess.row_work_csr(locals_[0], locals_[d].row_ptr)
        _all_reduce_sum(w, group)
        works.append(w)
    # ---- exchange: sizes of all shards in one tiny all-gather (host sync #1), then the all-gathers
    sizes = _exchange_sizes(locals_, group)
    wholes = [_gather_sampled(locals_[d], sizes[d], n_rows_global, group) for d in range(n_ds)]
    bounds_all = [sess.partition(works[d], n_ranks) for d in range(n_ds)]   # host sync #2 (one small D2H per event type, GPU idle-free: queued behind the gathers)
    # ---- compute phase: a rank transposes and expands only the item range it owns
    a = wholes[0]
    out: List[DevIndicators] = []
    for d in range(n_ds):
        bounds = bounds_all[d]
        a_col_ptr, a_row_idx = sess.transpose(a, counts[0], bounds[rank], bounds[rank + 1])
        out.append(sess.cco_rows(bounds[rank], bounds[rank + 1], n_items_a, a_col_ptr, a_row_idx, a.nnz_bound, wholes[d], counts[0], counts[d],
                                 n_rows_global, d == 0, params[d]))
    return ShardedResult(out, bounds_all, [w.nnz_bound for w in wholes])


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
