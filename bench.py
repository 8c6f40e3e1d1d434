#!/usr/bin/env python3
"""bench.py -- CCO model-build throughput on MI355X (BASELINE.json metric: cooccurrence pairs/sec (A'A + A'B) + LLR
top-k items/sec).

A "step" = one complete pass of the hot path over the workload with the raw per-event user x item CSR matrices
already resident in HBM, through the C ABI's context-level entry point `urcco_context_build_device` (include/urcco.h):
column counts -> sampleDownAndBinarize -> A.t -> per event type A.t %*% B fused with LLR + top-k (+ the RCCL
all-reduces / all-gather-v inside the library for N > 1).  Nothing is cached between steps except scratch buffers.

Workload at EVERY N (`--workload auto`): BASELINE config 4 (synthetic 10M users x 2M items, Zipf-1.0, 5 event types) --
the HBM-resident job the 1 -> 8 GPU curve is quoted on, STRONG-scaled: the same job on N GPUs, rank r generates and owns
users [r, r + 1) * 10M / N; items are range-partitioned by work inside the library.  A SCALE division of an N-GPU value by
the N = 1 value is therefore apples to apples (`n1_same_workload`).  configs[1] (the 30-user MovieLens sample, 35K pairs)
and configs[0] are parity-test cases, they cannot load a GPU.  `--workload config3|config5` run the other synthetic
configurations; at N = 1 the default line also carries a `config3` object (BASELINE config 3: the cache-resident 1M x 200K
job rounds 1 and 2 were quoted on) for continuity.

  python bench.py                                   N = 1, config 4
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N      one process per GPU (RCCL via unique id)
  python bench.py --gpus N --single-process          ONE process drives N GPUs (what a JVM host does: ncclCommInitAll)

value = cooccurrence pairs formed per second, whole job (all ranks), max-over-ranks time.  One JSON line on rank 0.
Extra objects on the N = 1 line: roofline (+ roofline_lds / _llr / _pcie / _valu), cpu_baseline (the C oracle on all host
cores, same workload), host_level (the PCIe-inclusive one-shot entry point a JNI shim binds), csr_row_scan_hbm_resident,
config3 (+ its ingest_to_model and scipy legs).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
PCIE_PEAK_GBS = 63.0   # same guide: PCIe gen5 x16, per direction
# LDS atomic issue peak used for the secondary roofline of the SpGEMM classes: ds_write_b32-class operations retire 64 B
# per clock per CU (MI355X_MICROARCH.md, LDS table) = 16 lanes / clk / CU; 256 CUs at 2.4 GHz
LDS_ATOMIC_PEAK_GOPS = 16 * 256 * 2.4

BIN_STAGES = ["cco_rows_micro", "cco_rows_wave", "cco_rows_block_small", "cco_rows_block", "cco_rows_cu_half", "cco_rows_cu", "cco_rows_global"]
# (round 6: the row kernels' template arguments end in DBG -- false in production -- and PK -- true: the instantiation for a B' whose words carry the
#  columns' counts, the one that runs on the bench workloads; the micro class is three kernels, 64 / 32 / 16 lanes per row; the flags kernel's second
#  argument is the 32-bit RNG)
STAGE_TO_KERNEL = {"cco_rows_micro": "cco_rows_micro_kernel<64, false, true>", "cco_rows_wave": "cco_rows_kernel<64, 1024, 2, false, false, true>",
                   "cco_rows_block_small": "cco_rows_kernel<256, 4096, 2, false, false, true>", "cco_rows_block": "cco_rows_kernel<256, 8192, 2, false, false, true>",
                   "cco_rows_cu_half": "cco_rows_kernel<512, 16384, 1, false, false, true>", "cco_rows_cu": "cco_rows_kernel<1024, 32768, 1, false, false, true>",
                   "cco_rows_global": "cco_rows_kernel<1024, 32768, 1, true, false, true>",
                   "downsample_flags": "downsample_flags_kernel<false, false>", "compact_indicators": "compact_indicators_kernel"}
NAMES = {"config3": "config3: synthetic 1M users x 200K items, Zipf-1.0, purchase/view/category-pref",
         "config4": "config4: synthetic 10M users x 2M items, Zipf-1.0, 5 event types (purchase/view/add-to-cart/search/category-pref)",
         "config5": "config5: synthetic 10M x 2M skewed (top 0.1 % of the items = 40 % of the interactions, 1 % heavy users x50), 5 event types, indicators form"}


def kernel_source_id() -> str:
    """Identity of the kernel sources the in-tree library is built from: sha256 over csrc/*, first 16 hex digits.  The PMC traffic file
    under profiles/ records the id it was collected on (tools/r05_measure.sh); a file with another id is not quoted."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "universal-recommender_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(csrc, name), "rb").read())
    return h.hexdigest()[:16]


def algorithmic_bytes(stage: str, f: dict) -> float:
    """Algorithmic HBM bytes of one launch of a stage (int32 indices, implicit values; L2-resident gathers
    of per-column counts are not counted).  f = per-event-type facts."""
    U, nnz, nnzs, IA, IB = f["n_users"], f["nnz_raw"], f["nnz_sampled"], f["n_items_a"], f["n_items_b"]
    if stage == "column_counts":
        return 4.0 * nnz + 4.0 * IB
    if stage == "downsample_flags":      # first half of the CSR row scan: read col_idx + row_ptr, write the keep bitmask
        return 4.0 * nnz + 8.0 * (U + 1) + nnz / 8.0
    if stage == "downsample_scan":
        return 3.0 * (nnz / 64.0) * 8.0
    if stage == "downsample_compact":    # second half: re-read col_idx + bitmask + prefix, write col_idx' and row_ptr'
        return 4.0 * nnz + nnz / 8.0 + nnz / 8.0 + 4.0 * nnzs + 16.0 * (U + 1)
    if stage == "csr_row_scan":          # SURVEY 8d "K2": what a single-pass scan would have to move
        return 8.0 * (U + 1) + 4.0 * nnz + 8.0 * (U + 1) + 4.0 * nnzs
    if stage == "transpose":
        return 4.0 * f["nnz_a"] + 8.0 * (U + 1) + 4.0 * f["nnz_a"] + 8.0 * (IA + 1) + 4.0 * IA
    if stage == "row_work":
        return 4.0 * f["nnz_a"] + 8.0 * (IA + 1) + 16.0 * f["nnz_a"] + 8.0 * IA
    if stage.startswith("cco_rows"):     # SURVEY 8d K4 + K5 restricted to the bin's rows
        b = BIN_STAGES.index(stage)
        rows, pairs, users, outs = f["bin_rows"][b], f["bin_pairs"][b], f["bin_users"][b], f["bin_out"][b]
        return 4.0 * rows + 16.0 * rows + 4.0 * users + 16.0 * users + 4.0 * pairs + 12.0 * outs + 4.0 * rows
    if stage == "entropy":               # per-item entropies + 16-bit counts, and (round 6) B' packed with its columns' counts: read and written once
        return 12.0 * IA + 6.0 * IB + 8.0 * nnzs
    if stage == "compact_indicators":
        return 16.0 * IA * 1.0 + 24.0 * f["nnz_out"]
    return 0.0


def host_level_leg(lib, data, n_users, seed, pairs_expected, reps=3):
    """The one-shot host-level entry point exactly as a JNI shim calls it (pageable host CSR in, indicator CSR out in the
    library's pinned pool): PCIe-inclusive wall time, plus the share of it during which the caller's arrays must stay pinned
    (urcco_cross_occurrence_stage: what a JNI critical section spans).  One warm-up call creates the process-wide context."""
    from universal_recommender_amd import _lib
    n = len(data)
    arr = (_lib.Dataset * n)()
    keep = []
    for d, (_, nc, rp, ci) in enumerate(data):
        rp = np.ascontiguousarray(rp, np.int64)
        ci = np.ascontiguousarray(ci, np.int32)
        keep += [rp, ci]
        arr[d].matrix.n_rows, arr[d].matrix.n_cols = n_users, nc
        arr[d].matrix.row_ptr, arr[d].matrix.col_idx = rp.ctypes.data, ci.ctypes.data
        arr[d].max_elements_per_row, arr[d].max_interesting_elements = 500, 50
    opts = _lib.Options(device=0, row_rate_mode=0, n_gpus=1)
    times, stage_times, pairs, nnz_out = [], [], 0, 0
    for it in range(reps + 1):
        out = (_lib.Indicators * n)()
        stats = (_lib.DatasetStats * n)()
        t0 = time.perf_counter()
        _lib.check(lib.urcco_cross_occurrence_stage(arr, n, seed, C.byref(opts)), lib)
        t1 = time.perf_counter()
        _lib.check(lib.urcco_cross_occurrence_finish(out, n, stats), lib)
        dt = time.perf_counter() - t0
        pairs = sum(int(s.pairs) for s in stats)
        nnz_out = sum(int(o.nnz) for o in out)
        lib.urcco_free_indicators(out, n)
        if it > 0:
            times.append(dt)
            stage_times.append(t1 - t0)
    lib.urcco_shutdown()
    ms = statistics.median(times) * 1e3
    h2d = sum(r.nbytes + c.nbytes for r, c in zip(keep[0::2], keep[1::2]))
    d2h = 12 * nnz_out + 8 * sum(data[0][1] + 1 for _ in data)
    # PCIe is full duplex: the link-bound time of the call is the longer direction at 63 GB/s
    bound_ms = max(h2d, d2h) / (PCIE_PEAK_GBS * 1e9) * 1e3
    return {"entry_point": "urcco_cross_occurrence_stage + _finish == urcco_cross_occurrence_downsampled (host CSR in pageable memory -> host indicator CSR)",
            "ms": round(ms, 3), "caller_arrays_pinned_ms": round(statistics.median(stage_times) * 1e3, 3),
            "pairs_per_s": round(pairs / (ms / 1e3), 1), "h2d_MB": round(h2d / 1e6, 1), "d2h_MB": round(d2h / 1e6, 1),
            "runs": f"median of {reps} after 1 warm-up (the warm-up creates the persistent context, its pinned staging and output pool)",
            "pairs_match": bool(pairs == pairs_expected),
            "roofline_pcie": {"bound": "pcie", "peak": PCIE_PEAK_GBS, "unit": "GB/s per direction", "h2d_GBps": round(h2d / 1e9 / (ms / 1e3), 2),
                              "d2h_GBps": round(d2h / 1e9 / (ms / 1e3), 2), "link_bound_ms": round(bound_ms, 3), "frac": round(bound_ms / ms, 4),
                              "note": "frac = (longer direction at the PCIe peak) / measured wall time of the whole call (H2D of pageable memory through the pinned "
                                      "ring, build, D2H); the directions overlap only at the end of the call"}}


def rowscan_hbm_leg(lib, dev, seed):
    """The CSR row scan (sampleDownAndBinarize: flags + scan + compact) on a matrix far beyond the 256 MiB Infinity Cache:
    config 3's `view` generator with 8M users (> 1 GiB of column indices), generated on the device."""
    from universal_recommender_amd import synth
    from universal_recommender_amd.device import DevCsr, DeviceSession
    cfg = synth.config3(1.0)
    cfg.n_users = 8_000_000
    cfg.events = [cfg.events[1]]
    cfg.events[0].lam = 39
    (_, nc, rp, ci), = synth.generate_device(cfg, dev)
    nnz = int(rp[-1].item())
    m = DevCsr(cfg.n_users, nc, rp, ci, nnz)
    sess = DeviceSession(dev, lib)
    raw = sess.column_counts(m.col_idx, nnz, nc)
    for _ in range(2):
        out, _post = sess.downsample(m, nnz, raw, seed, 500)
    torch.cuda.synchronize(dev)
    reps = 5
    sess.set_timing(True)
    for _ in range(reps):
        out, _post = sess.downsample(m, nnz, raw, seed, 500)
    tm = sess.get_timings()
    sess.set_timing(False)
    kept = int(out.row_ptr[-1].item())
    parts = {k: tm[k][0] / reps for k in ("downsample_flags", "downsample_scan", "downsample_compact") if tm[k][1]}
    ms = sum(parts.values())
    alg = 16.0 * (cfg.n_users + 1) + 4.0 * nnz + 4.0 * kept
    sess.close()
    return {"matrix": f"config 3 `view` generator, {cfg.n_users} users x {nc} items: nnz {nnz} ({4 * nnz / 2**30:.2f} GiB of column indices) -> {kept} kept",
            "ms": round(ms, 4), "parts_ms": {k: round(v, 4) for k, v in parts.items()}, "alg_MB": round(alg / 1e6, 1),
            "GBps": round(alg / 1e9 / (ms / 1e3), 1), "frac_of_hbm_peak": round(alg / 1e9 / (ms / 1e3) / HBM_PEAK_GBS, 4),
            "label": "HBM-resident (working set >> 256 MiB Infinity Cache)"}


def llr_rate_leg(lib, dev):
    """fp64 LLR evaluation rate of the device (SURVEY 8d secondary bound): the test hook urcco_dev_llr evaluates
    SimilarityAnalysis.logLikelihoodRatio in full (11 xLogX, i.e. up to 11 fdlibm-style logarithms) for n tuples."""
    from universal_recommender_amd.device import DeviceSession
    sess = DeviceSession(dev, lib)
    n = 1 << 24
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    n_users = torch.full((n,), 10_000_000, dtype=torch.int64, device=dev)
    a = torch.randint(1, 5000, (n,), device=dev, generator=g, dtype=torch.int64)
    b = torch.randint(1, 5000, (n,), device=dev, generator=g, dtype=torch.int64)
    ab = torch.minimum(torch.minimum(a, b), torch.randint(1, 50, (n,), device=dev, generator=g, dtype=torch.int64))
    sess.llr(a, b, ab, n_users)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(5):
        sess.llr(a, b, ab, n_users)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / 5
    sess.close()
    return n / dt


def ingest_leg(lib, dev, data, cfg, seed):
    """end-to-end events/s INCLUDING the copies -- the event streams of config 3 (every interaction once, in
    random stream order, plus ~10 % repeated events; 64-bit keys as the host's string hashing produces them) start in
    pageable host memory, go through the device Preparator (dictionaries by first appearance, CSR builders) and straight
    into the CCO build; the timed region ends when the indicator rows are back on the host."""
    from universal_recommender_amd import ingest
    from universal_recommender_amd.device import DatasetParams, DeviceSession, cross_occurrence_device
    rng = np.random.default_rng(1)
    host_actions, total = [], 0
    for (name, n_cols, rp, ci) in data:
        rows = np.repeat(np.arange(cfg.n_users, dtype=np.int64), np.diff(rp))
        cols = ci.astype(np.int64)
        dup = rng.integers(0, rows.size, rows.size // 10)
        rows, cols = np.concatenate([rows, rows[dup]]), np.concatenate([cols, cols[dup]])
        order = rng.permutation(rows.size)
        uk = rows[order] * np.int64(0x9E3779B97F4A7C15 - (1 << 64)) + 11      # injective stand-ins for 64-bit string hashes
        ik = cols[order] * np.int64(0xC2B2AE3D27D4EB4F - (1 << 64)) + 5
        host_actions.append((name, uk, ik))
        total += rows.size
    sess = DeviceSession(dev, lib)
    params = [DatasetParams(500, 50, None)] * len(data)
    times, pairs, parts = [], 0, None
    for it in range(3):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        actions = [(n, torch.from_numpy(u).to(dev), torch.from_numpy(i).to(dev)) for (n, u, i) in host_actions]
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        dp = ingest.prepare_device(sess, actions, 1)
        sess.synchronize()
        t2 = time.perf_counter()
        res = cross_occurrence_device(sess, [ev.matrix for ev in dp.events], params, seed)
        host = [r.to_host() for r in res]
        t3 = time.perf_counter()
        pairs = sum(int(r.stats[0]) for r in res)
        if it > 0:
            times.append(t3 - t0)
            parts = {"h2d_ms": round((t1 - t0) * 1e3, 2), "ingest_ms": round((t2 - t1) * 1e3, 2), "model_build_and_d2h_ms": round((t3 - t2) * 1e3, 2)}
        del actions, dp, res, host
    sess.close()
    s = statistics.median(times)
    return {"events": total, "ms": round(s * 1e3, 2), "events_per_s": round(total / s, 1), "parts_last_run": parts, "pairs": pairs,
            "what": "pageable host key streams (16 B per event) -> H2D -> device Preparator (2 dictionary builds + 2 lookups + CSR build per event type) -> "
                    "CCO model build -> indicator rows D2H; median of 2 after 1 warm-up; string hashing (urcco_hash_strings on the host) not included"}


def cpu_oracle_leg(data, n_users, seed, pairs_gpu, runs):
    """CPU baseline on this box's host cores: the C oracle (a restatement of the Mahout algorithm;
    Mahout/Spark local[*] itself cannot run here: no JVM, un-vendored jars) on every core, on the SAME workload."""
    from oracle import c_oracle as O
    cores = min(os.cpu_count() or 1, O.lib().orc_max_threads())
    mats = [O.Csr(n_users, nc, rp, ci) for (_, nc, rp, ci) in data]
    ps = [O.DatasetParams(500, 50, None)] * len(mats)
    times, cpu_pairs = [], 0
    for it in range(runs + (1 if runs > 1 else 0)):
        t0 = time.perf_counter()
        ref = O.cross_occurrence_downsampled(mats, ps, seed, 0, cores)
        dt = time.perf_counter() - t0
        cpu_pairs = sum(r.pairs for r in ref)
        if it > 0 or runs == 1:
            times.append(dt)
        del ref
    cpu_s = statistics.median(times)
    how = f"median of {runs} runs after 1 warm-up" if runs > 1 else "ONE run, no warm-up (the bounded sample: a run is tens of seconds of 100+ cores)"
    return {"value": round(cpu_pairs / cpu_s, 1), "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"the whole workload ({cpu_pairs} pairs), {how} ({cpu_s:.2f} s); C oracle, OpenMP (down-sampling, transposition and SpGEMM + LLR + top-k over "
                      "all cores); Mahout/Spark local[*] is not runnable in this image",
            "pairs_match_gpu": bool(cpu_pairs == pairs_gpu)}


def scipy_leg(data, n_users, seed):
    """Independent second CPU number: scipy, one thread, A'A of the primary event type only."""
    from oracle import c_oracle as O
    import scipy.sparse as sp
    (_, nc, rp, ci) = data[0]
    m0 = O.Csr(n_users, nc, rp, ci)
    a = O.downsample(m0, O.column_counts(m0), seed, 500)
    A = sp.csr_matrix((np.ones(a.nnz, np.int32), a.col_idx, a.row_ptr), shape=(a.n_rows, a.n_cols))
    cnt = np.asarray(A.sum(axis=0)).ravel().astype(np.int64)
    pairs_aa = int((np.diff(a.row_ptr).astype(np.int64) ** 2).sum())
    t0 = time.perf_counter()
    K = (A.T @ A).tocsr()
    K.setdiag(0)
    K.eliminate_zeros()

    def xlx(x):
        x = x.astype(np.float64)
        return np.where(x > 0, x * np.log(np.maximum(x, 1.0)), 0.0)
    rows = np.repeat(np.arange(K.shape[0]), np.diff(K.indptr))
    k11 = K.data.astype(np.int64)
    k12, k21 = cnt[rows] - k11, cnt[K.indices] - k11
    k22 = a.n_rows - cnt[rows] - cnt[K.indices] + k11
    row_e = xlx(k11 + k12 + k21 + k22) - xlx(k11 + k12) - xlx(k21 + k22)
    col_e = xlx(k11 + k12 + k21 + k22) - xlx(k11 + k21) - xlx(k12 + k22)
    mat_e = xlx(k11 + k12 + k21 + k22) - xlx(k11) - xlx(k12) - xlx(k21) - xlx(k22)
    llr = np.where(row_e + col_e < mat_e, 0.0, 2.0 * (row_e + col_e - mat_e))
    kept = 0
    for i in np.nonzero(np.diff(K.indptr) > 50)[0]:        # rows beyond k: top-k by partition
        s, e = K.indptr[i], K.indptr[i + 1]
        kept += np.argpartition(-llr[s:e], 50)[:50].size
    sc_s = time.perf_counter() - t0
    return {"value": round(pairs_aa / sc_s, 1), "unit": "pairs/s", "cores": 1, "kind": "port",
            "sample": f"A'A of the primary event type only ({pairs_aa} pairs, {K.nnz} distinct cooccurrences), one run ({sc_s:.2f} s): "
                      "scipy.sparse A.T @ A + vectorised numpy LLR + argpartition top-k, single thread"}


EMU_PHASES = [("input (counts + row scan on the user shard)", ["column_counts", "downsample_flags", "downsample_scan", "downsample_compact"]),
              ("own-shard transposition + fragment merge", ["transpose"]),
              ("exchange side (row lengths, need masks, masked lengths, packing, row_ptr rebuild)", ["exchange"]),
              ("expand preparation (row work)", ["row_work"]),
              ("binning + entropies", ["binning", "entropy"]),
              ("SpGEMM + LLR + top-k (item range)", BIN_STAGES),
              ("indicator compaction", ["compact_indicators"])]


def emulate_ranks_leg(library, dev, workload, W, args):
    """Per-rank critical path of a W-rank build measured on ONE GPU (readiness, not a scaling curve).  The W ranks of the
    job live in this process on this GPU (URCCO_FLAG_EMULATE_RANKS: one stream, every kernel alone on the device; collectives = device-to-
    device copies through DeviceLoopbackCollectives), the build is the real W-rank build -- user-range input phase, work-balanced item
    ranges from partition_dev, own-shard transposition + fragment merge, row-filtered all-to-all-v, fused expand, item-range SpGEMM --
    and every rank's HIP-event stage times are read separately.  What it cannot see: xGMI time, host enqueue time, overlap between streams."""
    from universal_recommender_amd import _lib, sharded, synth
    from universal_recommender_amd.device import Context, DatasetParams, DevCsr
    cfg = {"config3": synth.config3, "config4": synth.config4, "config5": synth.config5}[workload](args.scale)
    cuts = [cfg.n_users * g // W for g in range(W + 1)]
    per_rank = [synth.generate_device(cfg, dev, cuts[g], cuts[g + 1]) for g in range(W)]
    shards = [[DevCsr(cuts[g + 1] - cuts[g], per_rank[g][d][1], per_rank[g][d][2], per_rank[g][d][3], int(per_rank[g][d][2][-1].item())) for g in range(W)]
              for d in range(len(cfg.events))]
    params = [DatasetParams(500, 50, None) for _ in shards]
    builds = max(2, min(args.steps, 5))

    def run(n_ranks, sh, bases):
        coll = sharded.DeviceLoopbackCollectives(n_ranks, dev) if n_ranks > 1 else None
        flags = (_lib.FLAG_EMULATE_RANKS if n_ranks > 1 else _lib.FLAG_SINGLE_STREAM)
        ctx = Context(dev, library, n_ranks, flags, collectives=coll)
        try:
            ctx.build(sh, params, args.seed, cfg.n_users, bases)
            ctx.synchronize()
            ctx.set_timing(True)
            t0 = time.perf_counter()
            for _ in range(builds):
                ctx.build(sh, params, args.seed, cfg.n_users, bases)
            ctx.synchronize()
            wall = (time.perf_counter() - t0) / builds * 1e3
            tms = [ctx.get_timings_gpu(g) for g in range(n_ranks)]
            ctx.set_timing(False)
            res = ctx.results()
            pairs = [[int(ind.stats[0]) for ind in row] for row in res]          # [event type][rank]
            rows = [[ind.item_hi - ind.item_lo for ind in row] for row in res]
            if coll is not None and coll.error is not None:
                raise coll.error
            recv = [b / (builds + 1) for b in coll.bytes_received] if coll is not None else [0]
            return tms, pairs, rows, recv, wall
        finally:
            ctx.close()

    tms, pairs, rows, recv, wall_w = run(W, shards, cuts[:-1])
    # the same job on one rank (whole matrices = the shards' rows back to back), event types serialised on one stream: the N = 1 reference
    whole = []
    for d in range(len(cfg.events)):
        rp = [shards[d][0].row_ptr] + [shards[d][g].row_ptr[1:] + sum(shards[d][q].nnz_bound for q in range(g)) for g in range(1, W)]
        whole.append([DevCsr(cfg.n_users, shards[d][0].n_cols, torch.cat(rp), torch.cat([shards[d][g].col_idx[: shards[d][g].nnz_bound] for g in range(W)]),
                             sum(shards[d][g].nnz_bound for g in range(W)))])
    tms1, pairs1, _, _, wall_1 = run(1, whole, [0])
    assert sum(sum(p) for p in pairs) == sum(sum(p) for p in pairs1), "the W-rank build and the one-rank build form different numbers of pairs"

    def ms(t, names):
        return sum(t[n][0] for n in names if n in t) / builds
    phases = {}
    tot_rank = [0.0] * W
    for title, names in EMU_PHASES:
        per = [ms(t, names) for t in tms]
        for g in range(W):
            tot_rank[g] += per[g]
        mean = sum(per) / W
        phases[title] = {"per_rank_ms": [round(x, 3) for x in per], "max_ms": round(max(per), 3), "mean_ms": round(mean, 3),
                         "max_over_mean": round(max(per) / mean, 3) if mean > 0 else None, "one_rank_ms": round(ms(tms1[0], names), 3)}
    one = sum(v["one_rank_ms"] for v in phases.values())
    pr = [sum(pairs[d][g] for d in range(len(pairs))) for g in range(W)]
    out = {"what": f"EMULATED on one GPU, xGMI not included: the {W} ranks of a {workload} build run one after the other on this device (URCCO_FLAG_EMULATE_RANKS); "
                   "per-rank sums of HIP-event stage times, averaged over %d builds" % builds,
           "workload": workload, "scale": args.scale, "ranks": W, "phases": phases,
           "rank_total_ms": [round(x, 3) for x in tot_rank], "max_rank_ms": round(max(tot_rank), 3), "mean_rank_ms": round(sum(tot_rank) / W, 3),
           "max_over_mean": round(max(tot_rank) / (sum(tot_rank) / W), 3),
           "one_rank_serialised_ms": round(one, 3),
           "implied_compute_only_speedup": round(one / max(tot_rank), 3),
           "ideal_speedup_if_balanced": round(one / (sum(tot_rank) / W), 3),
           "pairs_per_rank": pr, "pairs_max_over_mean": round(max(pr) / (sum(pr) / W), 4),
           "item_rows_per_rank": [r for r in rows[0]],
           "bytes_received_per_rank_MB": [round(b / 1e6, 1) for b in recv],
           "xgmi_ms_at_300GBps_per_rank": [round(b / 300e9 * 1e3, 3) for b in recv],
           "wall_ms_per_build": {"emulated_W_ranks_sequential_incl_python_collectives": round(wall_w, 2), "one_rank_single_stream": round(wall_1, 2)}}
    return out


class Job:
    """One workload resident on this process's GPU(s) + its context: warm-up, timed region, the optional extra passes."""

    def __init__(self, library, workload, args, world, rank, devs, single_process):
        from universal_recommender_amd import _lib, sharded, synth
        from universal_recommender_amd.device import Context, DatasetParams, DevCsr
        self.lib, self.workload, self.args, self.world, self.rank, self.devs = library, workload, args, world, rank, devs
        self.single_process = single_process
        self.cfg = {"config3": synth.config3, "config4": synth.config4, "config5": synth.config5}[workload](args.scale)
        cfg = self.cfg
        n_local = len(devs)
        n_parts = world if not single_process else n_local
        first = rank if not single_process else 0
        self.cuts = [cfg.n_users * (first + g) // n_parts for g in range(n_local + 1)]
        t0 = time.time()
        self.host_data = None
        self.shards = []        # [event type][local gpu]
        if workload == "config3" and n_local == 1 and world == 1:
            self.host_data = synth.generate(cfg, self.cuts[0], self.cuts[1])
            self.shards = [[DevCsr(self.cuts[1] - self.cuts[0], nc, torch.from_numpy(rp).to(devs[0]), torch.from_numpy(ci).to(devs[0]), int(rp[-1]))]
                           for (_, nc, rp, ci) in self.host_data]
            self.generator = "numpy PCG64 on the host (universal_recommender_amd.synth.generate)"
        else:
            per_gpu = [synth.generate_device(cfg, devs[g], self.cuts[g], self.cuts[g + 1]) for g in range(n_local)]
            for d in range(len(cfg.events)):
                self.shards.append([DevCsr(self.cuts[g + 1] - self.cuts[g], per_gpu[g][d][1], per_gpu[g][d][2], per_gpu[g][d][3], int(per_gpu[g][d][2][-1].item()))
                                    for g in range(n_local)])
            self.generator = "torch Philox on the GPU (universal_recommender_amd.synth.generate_device)"
        for dv in devs:
            torch.cuda.synchronize(dv)
        self.gen_s = time.time() - t0
        self.params = [DatasetParams(500, 50, None) for _ in self.shards]   # engine.json defaults: maxEventsPerEventType 500, maxCorrelatorsPerEventType 50
        self.exchange = world > 1 or n_local > 1 or args.force_exchange
        self.base_flags = (_lib.FLAG_FORCE_EXCHANGE if args.force_exchange else 0)
        flags = self.base_flags | (_lib.FLAG_SINGLE_STREAM if args.single_stream else 0)
        # RCCL may greet on the C-level stdout when the first communicator comes up (version / host / library path): stdout carries
        # ONE JSON line, so file descriptor 1 points at stderr while the context is created
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if single_process:
                self.ctx = Context(devs[0], library, n_local, flags)      # ncclCommInitAll inside the library when n_local > 1
            else:
                self.ctx = sharded.make_context(devs[0], library, flags=flags)
        finally:
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
        if args.gathered_primary:
            self.ctx.set_debug(8192)
        if os.environ.get("URCCO_BENCH_DEBUG"):   # A/B aid: debug switches of the context (include/urcco.h, urcco_context_set_debug)
            self.ctx.set_debug(int(os.environ["URCCO_BENCH_DEBUG"]))

    def host_copy(self):
        """(name, n_cols, row_ptr, col_idx) numpy, whole matrices (single-GPU jobs)."""
        if self.host_data is None:
            self.host_data = [(ev.name, m[0].n_cols, m[0].row_ptr.cpu().numpy(), m[0].col_idx[: m[0].nnz_bound].cpu().numpy()) for ev, m in zip(self.cfg.events, self.shards)]
        return self.host_data

    def step(self):
        self.ctx.build(self.shards, self.params, self.args.seed, self.cfg.n_users, self.cuts[:-1])    # urcco_context_build_device: enqueue only

    def barrier(self):
        self.ctx.synchronize()
        if self.world > 1 and not self.single_process:
            dist.barrier()
        for dv in self.devs:
            torch.cuda.synchronize(dv)

    def timed(self, steps, warmup):
        for _ in range(warmup):
            self.step()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.barrier()
        return time.perf_counter() - t0

    def close(self):
        if self.ctx is not None:
            self.ctx.close()
            self.ctx = None
        self.shards = []


def measure(job: Job, args, full: bool):
    """Timed region + (full) single-build latency, unordered-rows pass, per-kernel table from a single-stream pass."""
    from universal_recommender_amd import _lib
    world, rank, single_process = job.world, job.rank, job.single_process
    ctx, cfg = job.ctx, job.cfg
    n_local = len(job.devs)
    for _ in range(args.warmup):
        job.step()
    job.barrier()
    if args.single_stream:
        ctx.set_timing(True)    # HIP events around every launch group, on the launching stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        job.step()
    job.barrier()
    elapsed = time.perf_counter() - t0
    if args.timed_only:
        return {"ms_per_step": round(elapsed / args.steps * 1e3, 4)}
    # one build at a time (what a single `pio train` sees): every build followed by a wait
    single_build_ms = None
    plain = world == 1 and n_local == 1 and not args.single_stream and not args.force_exchange
    if plain:
        lat = []
        for _ in range(max(5, args.steps // 2)):
            job.barrier()
            t1 = time.perf_counter()
            job.step()
            ctx.synchronize()
            lat.append((time.perf_counter() - t1) * 1e3)
        single_build_ms = statistics.median(lat)
    # the same steps with URCCO_FLAG_UNORDERED_ROWS (rows = top-k sets without the in-kernel ranking pass: what the JNI shim
    # asks for -- the Scala host re-inserts by column index anyway) -- reported beside `value`, never as `value`
    unordered = None
    if plain and full:
        ctx.set_flags(job.base_flags | _lib.FLAG_UNORDERED_ROWS)
        unordered = job.timed(args.steps, args.warmup) / args.steps
        ctx.set_flags(job.base_flags)
    if args.single_stream:
        timings = ctx.get_timings()
        ctx.set_timing(False)
        kernel_timing_mode = "timed region (one stream)"
    else:
        # The timed region overlaps the event types on separate HIP streams, so a kernel's event-bracketed duration there
        # includes time it shared the GPU with other kernels.  Per-kernel durations (roofline) are therefore taken from a
        # second pass of the same steps with the event types serialised on one stream (== `bench.py --single-stream`, the
        # command the rocprofv3 summary in profiles/ is taken from); every rank takes part.
        ctx.set_flags(job.base_flags | _lib.FLAG_SINGLE_STREAM)
        for _ in range(args.warmup):
            job.step()
        job.barrier()
        ctx.set_timing(True)
        for _ in range(args.steps):
            job.step()
        job.barrier()
        timings = ctx.get_timings()
        ctx.set_timing(False)
        ctx.set_flags(job.base_flags)
        kernel_timing_mode = f"separate single-stream pass of the same steps (the timed region overlaps the event types on {len(job.shards)} HIP streams)"
    if world > 1 and not single_process:
        t = torch.tensor([elapsed], dtype=torch.float64, device=job.devs[0])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- facts about the last step (identical every step: the build is a pure function of inputs + seed) ----
    res = ctx.results()                 # [event type][local gpu]
    dev0 = job.devs[0]
    stats_local = torch.stack([torch.stack([ind.stats.to(dev0) for ind in row]).sum(0) for row in res]).clone()
    stats = stats_local.clone()
    nnz_out = torch.tensor([sum(int(ind.row_ptr[-1]) for ind in row) for row in res], dtype=torch.int64, device=dev0)
    if world > 1 and not single_process:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        dist.all_reduce(nnz_out, op=dist.ReduceOp.SUM)
    stats = stats.cpu().numpy()
    stats_local = stats_local.cpu().numpy()
    nnz_out = nnz_out.cpu().numpy()
    inds = [row[0] for row in res]
    pairs_per_event = [int(s[0]) for s in stats]
    pairs = int(sum(pairs_per_event))
    n_items_a = cfg.events[0].n_items
    items = n_items_a * len(cfg.events)
    ms_per_step = elapsed / args.steps * 1e3
    value = pairs / (elapsed / args.steps)
    nnz_sampled = [ind.nnz_sampled_global() for ind in inds]  # the whole matrix over all ranks (the row-filtered exchange leaves a rank only its rows)
    nnz_raw_local = [sum(s.nnz_bound for s in row) for row in job.shards]
    item_range = [inds[0].item_lo, inds[0].item_hi]
    out = {"elapsed": elapsed, "ms_per_step": ms_per_step, "value": value, "pairs": pairs, "pairs_per_event": pairs_per_event, "stats": stats,
           "nnz_out": nnz_out, "nnz_sampled": nnz_sampled, "nnz_raw_local": nnz_raw_local, "item_range": item_range, "items": items,
           "single_build_ms": single_build_ms, "unordered": unordered, "kernel_timing_mode": kernel_timing_mode}
    if rank != 0:
        return out

    # ---- per-kernel table + roofline of the dominant kernel (this process's HIP-event timings, averaged over its GPUs) ----
    facts = []
    NB = _lib.N_BINS
    n_rows_local = (job.cuts[-1] - job.cuts[0]) // n_local
    gpus_in_table = n_local
    for d, ev in enumerate(cfg.events):
        st = stats_local[d] / gpus_in_table     # this process's own rows, per GPU
        facts.append(dict(n_users=n_rows_local, nnz_raw=nnz_raw_local[d] / gpus_in_table, nnz_sampled=nnz_sampled[d] // (world if not single_process else n_local),
                          nnz_a=nnz_sampled[0], n_items_a=n_items_a,
                          n_items_b=ev.n_items, k=50, bin_rows=[float(x) for x in st[1:1 + NB]], bin_pairs=[float(x) for x in st[1 + NB:1 + 2 * NB]],
                          bin_users=[float(x) for x in st[1 + 2 * NB:1 + 3 * NB]], bin_out=[float(x) for x in st[1 + 3 * NB:1 + 4 * NB]],
                          nnz_out=float(sum(int(ind.row_ptr[-1]) for ind in res[d])) / gpus_in_table))
    per_event_stages = ["column_counts", "downsample_flags", "downsample_scan", "downsample_compact", "row_work", "entropy"] + BIN_STAGES + ["compact_indicators"]
    kernels = {}
    for name, (ms, n) in timings.items():
        if n == 0:
            continue
        if name in per_event_stages:
            byts = sum(algorithmic_bytes(name, f) for f in facts)
        elif name == "transpose":
            byts = algorithmic_bytes(name, facts[0])
        else:
            byts = 0.0
        ms_step = ms / args.steps / gpus_in_table
        kernels[name] = {"ms_per_step": round(ms_step, 4), "launches_per_step": n // args.steps // gpus_in_table,
                         "alg_MB_per_step": round(byts / 1e6, 2), "GBps": round(byts / 1e9 / (ms_step / 1e3), 1) if ms_step > 0 and byts > 0 else None}
    scan_ms = sum(kernels[n]["ms_per_step"] for n in ("downsample_flags", "downsample_scan", "downsample_compact") if n in kernels)
    scan_bytes = sum(algorithmic_bytes("csr_row_scan", f) for f in facts)
    kernels["csr_row_scan(flags+scan+compact)"] = {"ms_per_step": round(scan_ms, 4), "alg_MB_per_step": round(scan_bytes / 1e6, 2),
                                                   "GBps": round(scan_bytes / 1e9 / (scan_ms / 1e3), 1) if scan_ms > 0 else None,
                                                   "frac_of_hbm_peak": round(scan_bytes / 1e9 / (scan_ms / 1e3) / HBM_PEAK_GBS, 4) if scan_ms > 0 else None,
                                                   "label": "cache-resident (the matrices fit the 256 MiB Infinity Cache)" if job.workload == "config3" else "HBM-resident"}
    spgemm_names = [n for n in BIN_STAGES if n in kernels]
    non_spgemm_ms = sum(v["ms_per_step"] for k, v in kernels.items() if k not in spgemm_names and not k.startswith("csr_row_scan"))
    timed = {k: v for k, v in kernels.items() if not k.startswith("csr_row_scan") and v["GBps"]}
    dominant = max(timed, key=lambda k: timed[k]["ms_per_step"])
    dk = timed[dominant]
    launches = max(dk["launches_per_step"], 1)
    # HBM traffic of the dominant kernel from the committed PMC passes (rocprofv3 cannot wrap the process it runs in).  The file must have
    # been collected on THESE kernel sources (kernel_source_id), else no traffic is quoted (round 3 quoted a stale one).
    traffic, traffic_src = None, None
    for tpath in (os.path.join(ROOT, "profiles", f"r06_hbm_traffic_pmc_{job.workload}.json"), os.path.join(ROOT, "profiles", f"r05_hbm_traffic_pmc_{job.workload}.json")):
        if traffic is not None:
            break
        if world == 1 and n_local == 1 and args.scale == 1.0 and os.path.exists(tpath) and dominant in STAGE_TO_KERNEL:
            tj = json.load(open(tpath))
            if tj.get("kernel_source_id") != kernel_source_id():
                traffic_src = f"{os.path.relpath(tpath, ROOT)} was collected on kernel sources {tj.get('kernel_source_id')}, this library is {kernel_source_id()}: not quoted"
                continue
            tk = tj["kernels"].get(STAGE_TO_KERNEL[dominant])
            if tk and "hbm_bytes_per_launch" in tk:
                traffic = tk["hbm_bytes_per_launch"]
                traffic_src = (os.path.relpath(tpath, ROOT) + f" (kernel sources {tj['kernel_source_id']}; ({tk['fetch_factor']} * FETCH_SIZE + WRITE_SIZE) KB -> bytes, two separate --pmc "
                               f"passes on this workload; FETCH_SIZE factor: {tk['fetch_factor_kind']}, profiles/r04_fetch_size_calibration.json)")
    roofline = {"bound": "hbm", "kernel": dominant, "achieved": dk["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(dk["GBps"] / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "bytes_per_launch": round(dk["alg_MB_per_step"] * 1e6 / launches), "avg_launch_ms": round(dk["ms_per_step"] / launches, 4)}
    # secondary roofline of the SpGEMM classes: LDS accumulate operations (>= one read + one atomic per pair) against the
    # LDS atomic issue peak -- these kernels are neither HBM- nor LDS-throughput bound but dependent-latency bound
    spgemm_ms = sum(kernels[n]["ms_per_step"] for n in spgemm_names)
    spgemm_pairs = sum(sum(f["bin_pairs"]) for f in facts)
    roofline_lds = None
    if spgemm_ms > 0:
        gops = 2.0 * spgemm_pairs / (spgemm_ms / 1e3) / 1e9
        roofline_lds = {"bound": "lds-atomic issue", "kernels": "cco_rows_* (the LDS accumulator classes)", "achieved": round(gops, 1), "peak": round(LDS_ATOMIC_PEAK_GOPS, 1),
                        "unit": "G lane-ops/s", "frac": round(gops / LDS_ATOMIC_PEAK_GOPS, 4), "ops_model": "2 LDS operations per cooccurrence pair (probe read + atomic add)",
                        "note": "far below both ceilings: per-row dependent chains (gather -> insert -> score -> select -> rank) at LDS-limited occupancy"}
    llr_ms = sum(kernels[n]["ms_per_step"] for n in BIN_STAGES + ["compact_indicators", "row_work"] if n in kernels)
    candidates = int(sum(int(s[30]) for s in stats_local)) // gpus_in_table   # distinct (i, j) scored per step (counted while timing is on)
    out.update({"kernels": kernels, "roofline": roofline, "roofline_lds": roofline_lds, "llr_ms": llr_ms, "spgemm_ms": spgemm_ms, "non_spgemm_ms": non_spgemm_ms,
                "candidates": candidates, "facts": facts})
    return out


def mark(what: str):
    """Progress on stderr (the JSON line is the only thing on stdout): locates a fault that leaves no other trace."""
    print(f"[bench {time.strftime('%H:%M:%S')}] {what}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="auto", choices=["auto", "config3", "config4", "config5"],
                    help="auto = BASELINE config 4 at every N (the workload the 1 -> 8 GPU curve is quoted on)")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the workload (debug only; the reported config says so)")
    ap.add_argument("--single-process", action="store_true", help="ONE process drives --gpus N GPUs (the JVM-shaped route: ncclCommInitAll inside the library)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the host-level, row-scan, LLR-rate and config-3 legs")
    ap.add_argument("--single-stream", action="store_true", help="run the event types back to back on one HIP stream")
    ap.add_argument("--force-exchange", action="store_true",
                    help="debug: run the N > 1 code path (RCCL collectives, CSC fragments of the primary) in a one-rank communicator")
    ap.add_argument("--gathered-primary", action="store_true",
                    help="A/B (N > 1 path): the primary's CSC from a pass of every rank over the whole gathered A' (rounds 1-2) instead of fragments")
    ap.add_argument("--timed-only", action="store_true", help="stop after the timed region (timeline captures)")
    ap.add_argument("--seed", type=int, default=20260925)
    ap.add_argument("--emulate-ranks", type=int, default=0,
                    help="measurement mode: the per-rank critical path of a W-rank build on ONE GPU (prints its own JSON object, not the bench line)")
    args = ap.parse_args()
    # No restart logic: round 3's supervisor (a child re-run after a death by signal) is gone with the fault it papered over -- a
    # missing barrier in the top-k select, cco_rows.hip "SHARE && T != WAVE" (DESIGN.md section 7).  A process that dies fails the run.
    try:  # a GPU fault aborts the process; with ~100 GB mapped a core dump alone would take ten minutes on a gpurun box
        import resource
        resource.setrlimit(resource.RLIMIT_CORE, (0, 0))
    except Exception:
        pass

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the CCO path has no CPU fallback")
    if args.single_process:
        if world > 1:
            raise SystemExit("--single-process is launched as a plain `python bench.py --gpus N --single-process`, not under torch.distributed.run")
        if torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"--gpus {args.gpus} --single-process: only {torch.cuda.device_count()} device(s) visible")
        devs = [torch.device("cuda", g) for g in range(args.gpus)]
        world = args.gpus
    else:
        if world != args.gpus and world > 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
        if args.gpus > 1 and world == 1:
            raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N "
                             "(or add --single-process to drive the N GPUs from this one process)")
        devs = [torch.device("cuda", local_rank)]
    torch.cuda.set_device(devs[0])
    dev = devs[0]
    if world > 1 and not args.single_process:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", device_id=dev)     # torch's group: unique-id broadcast, barriers, max-over-ranks time

    from universal_recommender_amd import _lib
    if not os.path.exists(_lib.DEFAULT_PATH):   # the in-tree HIP library normally travels with the repo; build it otherwise
        if local_rank == 0:
            import __graft_entry__
            __graft_entry__.build_hip()
        if world > 1 and not args.single_process:
            dist.barrier()
    library = _lib.load(os.environ.get("URCCO_LIB", _lib.DEFAULT_PATH))   # URCCO_LIB: A/B runs of two builds on one box
    if args.emulate_ranks > 1:
        wl = "config4" if args.workload == "auto" else args.workload
        print(json.dumps(emulate_ranks_leg(library, dev, wl, args.emulate_ranks, args)), flush=True)
        return

    workload = args.workload if args.workload != "auto" else "config4"
    mark(f"generating {workload} on the device")
    job = Job(library, workload, args, world, rank, devs, args.single_process)
    mark("inputs resident; timed region + per-stage pass")
    m = measure(job, args, full=True)
    mark("measured")
    if args.timed_only:
        print(json.dumps(m), flush=True)
        job.close()        # a clean teardown: profilers wrapping this process wait for every queue to drain
        if world > 1 and not args.single_process:
            dist.destroy_process_group()
        return
    if rank != 0:
        job.close()
        dist.destroy_process_group()
        return
    cfg = job.cfg
    pairs = m["pairs"]
    n1 = world == 1 and len(devs) == 1
    main_run = n1 and args.scale == 1.0 and not args.single_stream and not args.force_exchange

    extras = {}
    cpu_baseline = None
    roofline_llr = None
    if main_run:
        host = job.host_copy() if (not args.no_extras or not args.no_cpu_baseline) else None
        job.close()          # frees the context's ~45 GB before the other legs
        if not args.no_extras:
            mark("host-level leg")
            extras["host_level"] = host_level_leg(library, host, cfg.n_users, args.seed, pairs)
            extras["roofline_pcie"] = extras["host_level"].pop("roofline_pcie")
            mark("row-scan leg")
            extras["csr_row_scan_hbm_resident"] = rowscan_hbm_leg(library, dev, args.seed)
            mark("LLR-rate leg")
            rate = llr_rate_leg(library, dev)
            if m["spgemm_ms"] > 0 and m["candidates"] > 0:
                ach = m["candidates"] / (m["spgemm_ms"] / 1e3)
                roofline_llr = {"bound": "fp64 LLR evaluation", "achieved": round(ach / 1e9, 2), "peak": round(rate / 1e9, 2), "unit": "G LLR/s", "frac": round(ach / rate, 4),
                                "candidates_per_step": m["candidates"],
                                "note": "peak = measured rate of the full logLikelihoodRatio (11 xLogX with fdlibm-style logarithms, urcco_dev_llr) on 16M tuples; the kernels' "
                                        "own evaluation is cheaper (per-item entropies hoisted, xLogX from two tables): the SpGEMM classes are far from fp64-bound"}
        if not args.no_cpu_baseline:
            mark("cpu_baseline (C oracle on the host cores)")
            cpu_baseline = cpu_oracle_leg(host, cfg.n_users, args.seed, pairs, runs=5)   # median of >= 5 runs after a warm-up (~14 s each on config 4)
        del host
        job.host_data = None
        if not args.no_extras and workload != "config3":
            # ---- continuity: BASELINE config 3 (what rounds 1 and 2 were quoted on) as an extra object, same code, same process
            mark("config-3 object")
            a3 = argparse.Namespace(**vars(args))
            a3.steps, a3.warmup = max(args.steps, 20), max(args.warmup, 5)
            j3 = Job(library, "config3", a3, 1, 0, devs, False)
            m3 = measure(j3, a3, full=True)
            h3 = j3.host_copy()
            j3.close()
            k3 = m3["kernels"]
            c3 = {"workload": NAMES["config3"], "value": round(m3["value"], 1), "unit": "pairs/s", "ms_per_step": round(m3["ms_per_step"], 4), "steps": a3.steps, "warmup": a3.warmup,
                  "pairs_per_step": m3["pairs"], "single_build_latency_ms": round(m3["single_build_ms"], 4),
                  "unordered_rows_ms_per_step": round(m3["unordered"] * 1e3, 4), "roofline": m3["roofline"],
                  "serialised_ms": {"spgemm_llr_topk": round(m3["spgemm_ms"], 4), "everything_else": round(m3["non_spgemm_ms"], 4)},
                  "kernels_ms_per_step": {k: v["ms_per_step"] for k, v in k3.items()},
                  "csr_row_scan": k3["csr_row_scan(flags+scan+compact)"],
                  "ingest_to_model": ingest_leg(library, dev, h3, j3.cfg, args.seed)}
            if not args.no_cpu_baseline:
                c3["cpu_baseline_scipy"] = scipy_leg(h3, j3.cfg.n_users, args.seed)
            extras["config3"] = c3
    # measured VALU issue ceiling (tools/valu_microbench.py, committed under profiles/) beside the SQ counters of the dominant class
    # (the counters must have been collected on THESE kernel sources, like the traffic figure)
    roofline_valu = None
    vpath, spath = os.path.join(ROOT, "profiles", "r03_valu_microbench.json"), os.path.join(ROOT, "profiles", f"r06_sq_counters_pmc_{workload}.json")
    if n1 and os.path.exists(vpath) and os.path.exists(spath) and m["roofline"]["kernel"] in STAGE_TO_KERNEL:
        vb = json.load(open(vpath))
        sj = json.load(open(spath))
        sq = sj["kernels"].get(STAGE_TO_KERNEL[m["roofline"]["kernel"]]) if sj.get("kernel_source_id") == kernel_source_id() else None
        if sq and "SQ_INSTS_VALU" in sq:
            launch_s = m["roofline"]["avg_launch_ms"] / 1e3
            ach = sq["SQ_INSTS_VALU"] / launch_s / 1e9
            slow = max(r["G_wave_ops_per_s"] for r in vb["rows"] if r["kernel"] == "k_bfe_ilp")  # shifts / bit-field / select class: ~4.25 cycles per wave64 instruction
            roofline_valu = {"bound": "VALU issue", "kernel": m["roofline"]["kernel"], "achieved": round(ach, 1), "peak": vb["wave_valu_instructions_per_s_G"], "unit": "G wave-instructions/s",
                             "frac": round(ach / vb["wave_valu_instructions_per_s_G"], 4), "cycles_per_wave64_valu": vb["cycles_per_wave64_valu"],
                             "peak_shift_class": slow, "frac_of_shift_class_peak": round(ach / slow, 4),
                             "note": "peak = measured rate of independent 32-bit adds at 8 waves per SIMD (2.5 cycles per wave64 instruction); shifts, bit-field and select instructions "
                                     "issue at 4.25 cycles: a real mix lies between the two fractions",
                             "sources": [os.path.relpath(vpath, ROOT), os.path.relpath(spath, ROOT)]}

    NB = _lib.N_BINS
    stats = m["stats"]
    line = {
        "metric": "cooccurrence pairs/sec (A'A+A'B) + LLR top-k items/sec", "value": round(m["value"], 1), "unit": "pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(m["ms_per_step"], 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "int32 counts / f64 LLR", "data": "synthetic",
        "n1_same_workload": True,
        "config": {"workload": NAMES[workload] + ("" if args.scale == 1.0 else f" SCALED x{args.scale} (debug)"),
                   "n_users": cfg.n_users, "n_items": [ev.n_items for ev in cfg.events], "events": [ev.name for ev in cfg.events],
                   "generator": job.generator, "nnz_raw_rank0": m["nnz_raw_local"], "nnz_sampled": m["nnz_sampled"], "pairs_per_event": m["pairs_per_event"],
                   "maxEventsPerEventType": 500, "maxCorrelatorsPerEventType": 50, "seed": args.seed, "rank0_item_range": m["item_range"],
                   "entry_point": "urcco_context_build_device (C ABI, include/urcco.h)",
                   "launch": ("one process drives all GPUs (ncclCommInitAll inside the library)" if args.single_process and world > 1 else "one process per GPU"),
                   "parallelism": f"items range-partitioned over {world} GPU(s)" + (", RCCL inside the library: 2 all-reduces + 1 all-gather-v per event type, 1 all-reduce of the work key" if job.exchange else "")
                                  + ("" if args.single_stream else ", one HIP stream per event type")},
        "pairs_per_step": pairs, "items_per_sec": round(m["items"] / (m["llr_ms"] / 1e3), 1) if m["llr_ms"] > 0 else None,
        "items_per_sec_note": "sum over event types of nItems(A) / time of the SpGEMM+LLR+top-k stages (rank 0)",
        "indicator_entries": int(m["nnz_out"].sum()), "rows_by_accumulator": dict(zip(["micro", "wave", "block_small", "block", "cu_half", "cu", "multipass"], [int(sum(s[1 + b] for s in stats)) for b in range(NB)])),
        "roofline": m["roofline"], "roofline_lds": m["roofline_lds"], "roofline_llr": roofline_llr, "roofline_valu": roofline_valu,
        "serialised_ms": {"spgemm_llr_topk": round(m["spgemm_ms"], 4), "everything_else": round(m["non_spgemm_ms"], 4)},
        "kernels": m["kernels"], "kernel_timing": m["kernel_timing_mode"], "cpu_baseline": cpu_baseline,
        "gpu_over_cpu": round(m["value"] / cpu_baseline["value"], 1) if cpu_baseline else None,
        "input_generation_s": round(job.gen_s, 1),
        "single_build_latency_ms": None if m["single_build_ms"] is None else round(m["single_build_ms"], 4),
        "single_build_latency_note": "median wall time of one build followed by a wait (the timed region enqueues its builds back to back: consecutive builds overlap)",
        "unordered_rows": None if m["unordered"] is None else {"flag": "URCCO_FLAG_UNORDERED_ROWS", "ms_per_step": round(m["unordered"] * 1e3, 4),
                                                                 "pairs_per_s": round(pairs / m["unordered"], 1),
                                                                 "note": "same build, indicator rows as unordered top-k sets (no ranking pass): what the JNI shim asks for; not the headline value"},
    }
    line.update(extras)
    print(json.dumps(line), flush=True)   # flushed before any teardown: a process that dies later (abort() does not flush stdio) has still reported
    mark("line printed; tearing down")
    job.close()
    if world > 1 and not args.single_process:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
