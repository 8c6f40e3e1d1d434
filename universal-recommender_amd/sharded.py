"""Multi-GPU CCO model build, one process per GPU (how bench.py is launched under torch.distributed.run).

The build itself -- user-range input phase, work-balanced item ranges, the exchange, the item-range compute phase
 -- lives in the library (csrc/urcco_context.hip, urcco_context_build_device) and talks RCCL directly, so
that a single JVM process reaches every GPU the same way.  This module only wires a process group to it:

  * on GPUs: rank 0 asks the library for an RCCL unique id, torch.distributed broadcasts the 128 bytes, every rank
    creates its urcco_context with (world_size, rank, id) -> ncclCommInitRank inside the library;
  * on the CPU test-suite (kernel sources on the test-only host simulator, `gloo` group): the context gets
    `TorchCollectives`, an implementation of the urcco_collectives callbacks on host memory through torch.distributed.

Collectives per model build: per event type 2 all-reduces (int32 column counts before / after sampling), 1 tiny
all-gather of (rows, nnz', long rows) records and an all-gather-v of row lengths (16-bit when they fit) + column indices
(no padding); for the primary also 1 all-reduce of the int64 row-work key, 1 tiny all-gather of fragment records and an
all-to-all-v of CSC fragments (every rank transposes its own user shard and sends each rank the columns of that rank's item
range: 16-bit column lengths + entries).  The host blocks once per event type on that event's own stream (shard sizes; for
the primary the same read brings the range bounds and the fragment sizes).  Mahout does the same job with Spark broadcasts of the count vectors and a shuffle inside
`A.t %*% B` (reference call sites URAlgorithm.scala:323-346).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .device import Context, DatasetParams, DevCsr, DevIndicators


@dataclass
class ShardedResult:
    indicators: List[DevIndicators]       # this rank's rows, one entry per event type
    item_ranges: List[List[int]]          # per event type: world_size + 1 bounds (filled by gather_item_ranges)
    nnz_sampled: List[int]                # global nnz after down-sampling, per event type


class TorchCollectives:
    """urcco_collectives on HOST memory through a torch.distributed group (the CPU test-suite's stand-in for RCCL).
    `local_ranks` = the global ranks this process holds (several when one process drives several simulated GPUs).  Calls
    are recorded between group_start and group_end and executed at group_end: the j-th call of every local rank is one
    collective."""

    def __init__(self, world: int, local_ranks: Sequence[int], group=None):
        self.world, self.local, self.group = world, list(local_ranks), group
        self.multi_process = dist.is_initialized() and dist.get_world_size(group) > 1
        self.pending = {}
        self.depth = 0
        self.error: Optional[BaseException] = None
        self.log: List[tuple] = []   # (kind, rank, bytes sent) of every collective executed (tests read the wire widths off it)
        self._gs = _lib.GROUP_FN(self._group_start)
        self._ge = _lib.GROUP_FN(self._group_end)
        self._ar = _lib.ALL_REDUCE_FN(self._all_reduce)
        self._ag = _lib.ALL_GATHER_V_FN(self._all_gather_v)
        self._aa = _lib.ALL_TO_ALL_V_FN(self._all_to_all_v)
        self.struct = _lib.Collectives(None, self._gs, self._ge, self._ar, self._ag, self._aa)

    def _group_start(self, user):
        self.depth += 1
        return 0

    def _all_reduce(self, user, rank, buf, count, dtype, stream):
        self.pending.setdefault(rank, []).append(("ar", buf, count, dtype))
        return 0

    def _all_gather_v(self, user, rank, send, recv, offsets, counts, stream):
        off = [offsets[r] for r in range(self.world)]
        cnt = [counts[r] for r in range(self.world)]
        self.pending.setdefault(rank, []).append(("ag", send, recv, off, cnt))
        return 0

    def _all_to_all_v(self, user, rank, send, soff, scnt, recv, roff, rcnt, stream):
        w = range(self.world)
        self.pending.setdefault(rank, []).append(("aa", send, [soff[r] for r in w], [scnt[r] for r in w], recv, [roff[r] for r in w], [rcnt[r] for r in w]))
        return 0

    def _group_end(self, user):
        self.depth -= 1
        if self.depth > 0:
            return 0
        try:
            n_ops = {len(v) for v in self.pending.values()}
            assert len(n_ops) <= 1 and set(self.pending) <= set(self.local), "local ranks issued different collective sequences"
            for j in range(n_ops.pop() if n_ops else 0):
                ops = {r: self.pending[r][j] for r in self.local}
                kind = ops[self.local[0]][0]
                assert all(op[0] == kind for op in ops.values()), "local ranks issued different collectives"
                if kind == "ar":
                    self._run_all_reduce(ops)
                elif kind == "ag":
                    self._run_all_gather_v(ops)
                else:
                    self._run_all_to_all_v(ops)
            self.pending = {}
            return 0
        except BaseException as e:  # surfaces as URCCO_RCCL_ERROR; the test reads .error
            self.error = e
            self.pending = {}
            return 1

    def _run_all_reduce(self, ops):
        views = []
        for r in self.local:
            _, buf, count, dtype = ops[r]
            self.log.append(("ar", r, count * (4 if dtype == 0 else 8)))
            ct = C.c_int32 if dtype == 0 else C.c_int64
            views.append(np.ctypeslib.as_array((ct * count).from_address(buf)))
        total = np.sum(views, axis=0, dtype=views[0].dtype)
        if self.multi_process:
            t = torch.from_numpy(total)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        for v in views:
            v[:] = total

    def _run_all_gather_v(self, ops):
        pieces = {}
        for r in self.local:
            _, send, recv, off, cnt = ops[r]
            self.log.append(("ag", r, cnt[r]))
            pieces[r] = bytes((C.c_char * cnt[r]).from_address(send)) if cnt[r] > 0 else b""
        if self.multi_process:
            gathered = [None] * dist.get_world_size(self.group)
            dist.all_gather_object(gathered, pieces, group=self.group)
            for g in gathered:
                pieces.update(g)
        for r in self.local:
            _, send, recv, off, cnt = ops[r]
            for p in range(self.world):
                assert len(pieces[p]) == cnt[p], f"rank {p} sent {len(pieces[p])} bytes, {cnt[p]} expected"
                if cnt[p] > 0:
                    C.memmove(recv + off[p], pieces[p], cnt[p])


    def _run_all_to_all_v(self, ops):
        pieces = {}  # (source, destination) -> bytes
        for r in self.local:
            _, send, soff, scnt, recv, roff, rcnt = ops[r]
            self.log.append(("aa", r, tuple(scnt)))
            for q in range(self.world):
                pieces[(r, q)] = bytes((C.c_char * scnt[q]).from_address(send + soff[q])) if scnt[q] > 0 else b""
        if self.multi_process:
            gathered = [None] * dist.get_world_size(self.group)
            dist.all_gather_object(gathered, pieces, group=self.group)
            for g in gathered:
                pieces.update(g)
        for r in self.local:
            _, send, soff, scnt, recv, roff, rcnt = ops[r]
            for p in range(self.world):
                assert len(pieces[(p, r)]) == rcnt[p], f"rank {p} sent {len(pieces[(p, r)])} bytes to rank {r}, {rcnt[p]} expected"
                if rcnt[p] > 0:
                    C.memmove(recv + roff[p], pieces[(p, r)], rcnt[p])


class DeviceLoopbackCollectives(TorchCollectives):
    """urcco_collectives for URCCO_FLAG_EMULATE_RANKS (bench.py --emulate-ranks, tests): every rank of the job lives in this process on
    ONE GPU, so a collective is a handful of device-to-device copies through zero-copy views on the library's buffers.  The device is
    drained before and after (the library enqueues on its own stream, torch on its own): measurement plumbing, not a data path."""

    def __init__(self, world: int, device):
        super().__init__(world, list(range(world)), None)
        self.device = torch.device(device)
        self.bytes_received = [0] * world   # per rank, over the builds so far: what xGMI would have carried to it

    def _group_end(self, user):
        if self.depth == 1 and self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        rc = super()._group_end(user)
        if self.depth == 0 and self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        return rc

    def _v(self, ptr, n, dtype=torch.uint8):
        from .device import _view
        return _view(ptr, n, dtype, self.device)

    def _run_all_reduce(self, ops):
        views = []
        for r in self.local:
            _, buf, count, dtype = ops[r]
            self.log.append(("ar", r, count * (4 if dtype == 0 else 8)))
            views.append(self._v(buf, count, torch.int32 if dtype == 0 else torch.int64))
        total = views[0].clone()
        for v in views[1:]:
            total += v
        for r, v in zip(self.local, views):
            v.copy_(total)
            self.bytes_received[r] += 2 * v.numel() * v.element_size() * (self.world - 1) // self.world   # reduce-scatter + all-gather

    def _run_all_gather_v(self, ops):
        for r in self.local:
            _, send, recv, off, cnt = ops[r]
            self.log.append(("ag", r, cnt[r]))
            for p in range(self.world):
                src = ops[p][1]
                if cnt[p] > 0 and recv + off[p] != src:
                    self._v(recv + off[p], cnt[p]).copy_(self._v(src, cnt[p]))
                if p != r:
                    self.bytes_received[r] += cnt[p]

    def _run_all_to_all_v(self, ops):
        for r in self.local:
            _, send, soff, scnt, recv, roff, rcnt = ops[r]
            self.log.append(("aa", r, tuple(scnt)))
            for p in range(self.world):
                _, send_p, soff_p, scnt_p, _, _, _ = ops[p]
                assert scnt_p[r] == rcnt[p], f"rank {p} sends {scnt_p[r]} bytes to rank {r}, {rcnt[p]} expected"
                if rcnt[p] > 0:
                    self._v(recv + roff[p], rcnt[p]).copy_(self._v(send_p + soff_p[r], rcnt[p]))
                if p != r:
                    self.bytes_received[r] += rcnt[p]


def make_context(device, library=None, group=None, flags: int = 0, row_rate_mode: int = _lib.ROW_RATE_MAHOUT_INT_DIV) -> Context:
    """The urcco_context of THIS rank of a one-process-per-GPU job (`group` = its torch.distributed group)."""
    device = torch.device(device)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    exchange = world > 1 or (flags & _lib.FLAG_FORCE_EXCHANGE)
    if not exchange:
        return Context(device, library, 1, flags, row_rate_mode)
    if device.type != "cuda":
        return Context(device, library, 1, flags, row_rate_mode, world, rank, None, TorchCollectives(world, [rank], group))
    box = [Context.unique_id(library) if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0, group=group)
    return Context(device, library, 1, flags, row_rate_mode, world, rank, box[0], None)


def cross_occurrence_sharded(ctx: Context, shards: Sequence[DevCsr], params: Sequence[DatasetParams], seed: int, n_rows_global: int,
                             row_base: int, wait: bool = True) -> ShardedResult:
    """SimilarityAnalysis.crossOccurrenceDownsampled over the job's GPUs.  shards[d] = this rank's user rows
    [row_base, row_base + n_rows) of event type d (shards[0] = primary).  Returns this rank's indicator rows as views
    on context-owned memory (valid until the context's next build)."""
    ctx.build([[m] for m in shards], params, seed, n_rows_global, [row_base])
    if not wait:
        return ShardedResult([], [], [])
    inds = [r[0] for r in ctx.results()]
    if ctx.collectives_error() is not None:
        raise ctx.collectives_error()
    return ShardedResult(inds, [], [i.nnz_sampled_global() for i in inds])


def gather_item_ranges(res: ShardedResult, n_items_a: int, group=None) -> List[int]:
    """world_size + 1 range bounds, identical on every rank (each rank contributes its own [lo, hi))."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    mine = (res.indicators[0].item_lo, res.indicators[0].item_hi)
    if world == 1:
        return [mine[0], mine[1]]
    parts = [None] * world
    dist.all_gather_object(parts, mine, group=group)
    assert parts[0][0] == 0 and parts[-1][1] == n_items_a and all(parts[r][1] == parts[r + 1][0] for r in range(world - 1)), parts
    return [p[0] for p in parts] + [parts[-1][1]]


def gather_indicators_to_host(res: ShardedResult, group=None):
    """Concatenate every rank's indicator rows on every rank (host numpy): list of (row_ptr, col_idx, llr).
    Not part of the timed model build (the reference hands the rows to URModel.save per partition)."""
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
