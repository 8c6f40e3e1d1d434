#!/usr/bin/env python3
"""bench.py -- CCO model-build throughput on MI355X (BASELINE.json metric: cooccurrence pairs/sec (A'A + A'B) + LLR
top-k items/sec).

A "step" = one complete pass of the hot path over the workload with the raw per-event user x item CSR matrices
already resident in HBM, through the C ABI's context-level entry point `urcco_context_build_device` (include/urcco.h):
column counts -> sampleDownAndBinarize -> A.t -> per event type A.t %*% B fused with LLR + top-k (+ the RCCL
all-reduces / all-gather-v inside the library for N > 1).  Nothing is cached between steps except scratch buffers.

  N = 1   workload = BASELINE config 3 (synthetic 1M users x 200K items, Zipf-1.0, purchase/view/category-pref) --
          configs[1] (the 30-user MovieLens sample, 35K pairs) is a parity-test case, it cannot load a GPU.
          `--workload config4|config5` runs the 10M x 2M jobs on one GPU (the N = 1 point of the strong-scaling curve).
  N > 1   STRONG scaling of BASELINE config 4 (10M users x 2M items, 5 event types) as named: the same job on N GPUs,
          rank r generates and owns users [r, r + 1) * 10M / N; items are range-partitioned by work inside the library.

value = cooccurrence pairs formed per second, whole job (all ranks), max-over-ranks time.  One JSON line on rank 0.
Extra objects on the N = 1 line: roofline (+ roofline_lds), cpu_baseline (median of 5 after a warm-up, all host cores)
and cpu_baseline_scipy (single thread), host_level (the PCIe-inclusive one-shot entry point a JNI shim binds),
csr_row_scan_hbm_resident (the row scan on a > 1 GiB matrix).
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
# LDS atomic issue peak used for the secondary roofline of the SpGEMM classes: ds_write_b32-class operations retire 64 B
# per clock per CU (MI355X_MICROARCH.md, LDS table) = 16 lanes / clk / CU; 256 CUs at 2.4 GHz
LDS_ATOMIC_PEAK_GOPS = 16 * 256 * 2.4

BIN_STAGES = ["cco_rows_micro", "cco_rows_wave", "cco_rows_block_small", "cco_rows_block", "cco_rows_cu_half", "cco_rows_cu", "cco_rows_global"]
STAGE_TO_KERNEL = {"cco_rows_micro": "cco_rows_micro_kernel", "cco_rows_wave": "cco_rows_kernel<64, 1024, 1>",
                   "cco_rows_block_small": "cco_rows_kernel<256, 4096, 1>", "cco_rows_block": "cco_rows_kernel<256, 8192, 1>",
                   "cco_rows_cu_half": "cco_rows_kernel<512, 16384, 1>", "cco_rows_cu": "cco_rows_kernel<1024, 32768, 1>",
                   "downsample_flags": "downsample_flags_kernel", "compact_indicators": "compact_indicators_kernel"}


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
    if stage == "compact_indicators":
        return 16.0 * IA * 1.0 + 24.0 * f["nnz_out"]
    return 0.0


def host_level_leg(lib, data, cfg, seed, pairs_expected, reps=3):
    """The one-shot host-level entry point exactly as a JNI shim calls it (pageable host CSR in, indicator CSR out in the
    library's pinned pool): PCIe-inclusive wall time.  One warm-up call creates the process-wide context."""
    from universal_recommender_amd import _lib
    n = len(data)
    arr = (_lib.Dataset * n)()
    keep = []
    for d, (_, nc, rp, ci) in enumerate(data):
        rp = np.ascontiguousarray(rp, np.int64)
        ci = np.ascontiguousarray(ci, np.int32)
        keep += [rp, ci]
        arr[d].matrix.n_rows, arr[d].matrix.n_cols = cfg.n_users, nc
        arr[d].matrix.row_ptr, arr[d].matrix.col_idx = rp.ctypes.data, ci.ctypes.data
        arr[d].max_elements_per_row, arr[d].max_interesting_elements = 500, 50
    opts = _lib.Options(device=0, row_rate_mode=0, n_gpus=1)
    times, pairs, nnz_out = [], 0, 0
    for it in range(reps + 1):
        out = (_lib.Indicators * n)()
        stats = (_lib.DatasetStats * n)()
        t0 = time.perf_counter()
        _lib.check(lib.urcco_cross_occurrence_downsampled(arr, n, seed, C.byref(opts), out, stats), lib)
        dt = time.perf_counter() - t0
        pairs = sum(int(s.pairs) for s in stats)
        nnz_out = sum(int(o.nnz) for o in out)
        lib.urcco_free_indicators(out, n)
        if it > 0:
            times.append(dt)
    lib.urcco_shutdown()
    ms = statistics.median(times) * 1e3
    h2d = sum(r.nbytes + c.nbytes for r, c in zip(keep[0::2], keep[1::2]))
    d2h = 12 * nnz_out + 8 * sum(data[0][1] + 1 for _ in data)
    return {"entry_point": "urcco_cross_occurrence_downsampled (host CSR in pageable memory -> host indicator CSR)", "ms": round(ms, 3),
            "pairs_per_s": round(pairs / (ms / 1e3), 1), "h2d_MB": round(h2d / 1e6, 1), "d2h_MB": round(d2h / 1e6, 1),
            "runs": f"median of {reps} after 1 warm-up (the warm-up creates the persistent context, its pinned staging and output pool)",
            "pairs_match": bool(pairs == pairs_expected)}


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
    parts = {k: tm[k][0] / reps for k in ("downsample_flags", "downsample_scan", "downsample_compact")}
    ms = sum(parts.values())
    alg = 16.0 * (cfg.n_users + 1) + 4.0 * nnz + 4.0 * kept
    sess.close()
    return {"matrix": f"config 3 `view` generator, {cfg.n_users} users x {nc} items: nnz {nnz} ({4 * nnz / 2**30:.2f} GiB of column indices) -> {kept} kept",
            "ms": round(ms, 4), "parts_ms": {k: round(v, 4) for k, v in parts.items()}, "alg_MB": round(alg / 1e6, 1),
            "GBps": round(alg / 1e9 / (ms / 1e3), 1), "frac_of_hbm_peak": round(alg / 1e9 / (ms / 1e3) / HBM_PEAK_GBS, 4),
            "label": "HBM-resident (working set >> 256 MiB Infinity Cache)"}


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


def cpu_legs(data, cfg, seed, pairs_gpu):
    """CPU baselines on this box's host cores: the C oracle (a restatement of the Mahout algorithm;
    Mahout/Spark local[*] itself cannot run here: no JVM, un-vendored jars) on every core, median of 5 after a warm-up;
    and single-thread scipy.sparse A.T @ B + numpy LLR + argpartition top-k on the primary event type."""
    from oracle import c_oracle as O
    cores = min(os.cpu_count() or 1, O.lib().orc_max_threads())
    mats = [O.Csr(cfg.n_users, nc, rp, ci) for (_, nc, rp, ci) in data]
    ps = [O.DatasetParams(500, 50, None)] * len(mats)
    times, cpu_pairs = [], 0
    for it in range(6):
        t0 = time.perf_counter()
        ref = O.cross_occurrence_downsampled(mats, ps, seed, 0, cores)
        dt = time.perf_counter() - t0
        cpu_pairs = sum(r.pairs for r in ref)
        if it > 0:
            times.append(dt)
        del ref
    cpu_s = statistics.median(times)
    base = {"value": round(cpu_pairs / cpu_s, 1), "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"the whole workload ({cpu_pairs} pairs), median of 5 runs after 1 warm-up ({cpu_s:.2f} s); C oracle, OpenMP "
                      "(down-sampling and SpGEMM + LLR + top-k over all cores); Mahout/Spark local[*] is not runnable in this image",
            "pairs_match_gpu": bool(cpu_pairs == pairs_gpu)}
    # ---- independent second number: scipy, one thread, A'A only
    import scipy.sparse as sp
    a = O.downsample(mats[0], O.column_counts(mats[0]), seed, 500)
    A = sp.csr_matrix((np.ones(a.nnz, np.int32), a.col_idx, a.row_ptr), shape=(a.n_rows, a.n_cols))
    cnt = np.asarray(A.sum(axis=0)).ravel().astype(np.int64)
    pairs_aa = int((np.diff(a.row_ptr).astype(np.int64) ** 2).sum())
    t0 = time.perf_counter()
    K = (A.T @ A).tocsr()
    K.setdiag(0)
    K.eliminate_zeros()
    n_users = a.n_rows

    def xlx(x):
        x = x.astype(np.float64)
        return np.where(x > 0, x * np.log(np.maximum(x, 1.0)), 0.0)
    rows = np.repeat(np.arange(K.shape[0]), np.diff(K.indptr))
    k11 = K.data.astype(np.int64)
    k12, k21 = cnt[rows] - k11, cnt[K.indices] - k11
    k22 = n_users - cnt[rows] - cnt[K.indices] + k11
    row_e = xlx(k11 + k12 + k21 + k22) - xlx(k11 + k12) - xlx(k21 + k22)
    col_e = xlx(k11 + k12 + k21 + k22) - xlx(k11 + k21) - xlx(k12 + k22)
    mat_e = xlx(k11 + k12 + k21 + k22) - xlx(k11) - xlx(k12) - xlx(k21) - xlx(k22)
    llr = np.where(row_e + col_e < mat_e, 0.0, 2.0 * (row_e + col_e - mat_e))
    kept = 0
    for i in np.nonzero(np.diff(K.indptr) > 50)[0]:        # rows beyond k: top-k by partition
        s, e = K.indptr[i], K.indptr[i + 1]
        kept += np.argpartition(-llr[s:e], 50)[:50].size
    sc_s = time.perf_counter() - t0
    scipy_leg = {"value": round(pairs_aa / sc_s, 1), "unit": "pairs/s", "cores": 1, "kind": "port",
                 "sample": f"A'A of the primary event type only ({pairs_aa} pairs, {K.nnz} distinct cooccurrences), one run ({sc_s:.2f} s): "
                           "scipy.sparse A.T @ A + vectorised numpy LLR + argpartition top-k, single thread"}
    return base, scipy_leg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="auto", choices=["auto", "config3", "config4", "config5"],
                    help="auto: config 3 at N = 1 (continuity with round 1), config 4 strong-scaled at N > 1")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the workload (debug only; the reported config says so)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the host-level and HBM-resident row-scan legs")
    ap.add_argument("--single-stream", action="store_true", help="run the event types back to back on one HIP stream")
    ap.add_argument("--force-exchange", action="store_true",
                    help="debug: run the N > 1 code path (RCCL collectives, range-restricted transpose) in a one-rank communicator")
    ap.add_argument("--timed-only", action="store_true", help="stop after the timed region (timeline captures)")
    ap.add_argument("--seed", type=int, default=20260925)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the CCO path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", device_id=dev)     # torch's group: unique-id broadcast, barriers, max-over-ranks time

    from universal_recommender_amd import _lib, sharded, synth
    from universal_recommender_amd.device import DatasetParams, DevCsr
    if not os.path.exists(_lib.DEFAULT_PATH):   # the in-tree HIP library normally travels with the repo; build it otherwise
        if local_rank == 0:
            import __graft_entry__
            __graft_entry__.build_hip()
        if world > 1:
            dist.barrier()
    library = _lib.load(os.environ.get("URCCO_LIB", _lib.DEFAULT_PATH))   # URCCO_LIB: A/B runs of two builds on one box

    # ---- workload -------------------------------------------------------------------------------------
    workload = args.workload if args.workload != "auto" else ("config3" if world == 1 else "config4")
    cfg = {"config3": synth.config3, "config4": synth.config4, "config5": synth.config5}[workload](args.scale)
    names = {"config3": "config3: synthetic 1M users x 200K items, Zipf-1.0, purchase/view/category-pref",
             "config4": "config4: synthetic 10M users x 2M items, Zipf-1.0, 5 event types (purchase/view/add-to-cart/search/category-pref)",
             "config5": "config5: synthetic 10M x 2M skewed (top 0.1 % of the items = 40 % of the interactions, 1 % heavy users x50), 5 event types, indicators form"}
    lo, hi = cfg.n_users * rank // world, cfg.n_users * (rank + 1) // world
    t0 = time.time()
    host_data = None
    if workload == "config3":
        host_data = synth.generate(cfg, lo, hi)
        shards = [DevCsr(hi - lo, nc, torch.from_numpy(rp).to(dev), torch.from_numpy(ci).to(dev), int(rp[-1])) for (_, nc, rp, ci) in host_data]
        generator = "numpy PCG64 on the host (universal_recommender_amd.synth.generate)"
    else:
        shards = [DevCsr(hi - lo, nc, rp, ci, int(rp[-1].item())) for (_, nc, rp, ci) in synth.generate_device(cfg, dev, lo, hi)]
        generator = "torch Philox on the GPU (universal_recommender_amd.synth.generate_device)"
    torch.cuda.synchronize(dev)
    gen_s = time.time() - t0
    params = [DatasetParams(500, 50, None) for _ in shards]   # engine.json defaults: maxEventsPerEventType 500, maxCorrelatorsPerEventType 50
    exchange = world > 1 or args.force_exchange
    base_flags = (_lib.FLAG_FORCE_EXCHANGE if args.force_exchange else 0)
    ctx = sharded.make_context(dev, library, flags=base_flags | (_lib.FLAG_SINGLE_STREAM if args.single_stream else 0))

    def step():
        ctx.build([[m] for m in shards], params, args.seed, cfg.n_users, [lo])    # urcco_context_build_device: enqueue only

    def barrier():
        ctx.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    if args.single_stream:
        ctx.set_timing(True)    # HIP events around every launch group, on the launching stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if args.timed_only:
        print(json.dumps({"ms_per_step": round(elapsed / args.steps * 1e3, 4)}))
        return
    # one build at a time (what a single `pio train` sees): every build followed by a wait
    single_build_ms = None
    if world == 1 and not args.single_stream and not args.force_exchange:
        lat = []
        for _ in range(max(5, args.steps // 2)):
            barrier()
            t1 = time.perf_counter()
            step()
            ctx.synchronize()
            lat.append((time.perf_counter() - t1) * 1e3)
        single_build_ms = statistics.median(lat)
    # the same steps with URCCO_FLAG_UNORDERED_ROWS (rows = top-k sets without the in-kernel ranking pass: what a JNI host,
    # which re-inserts by column index anyway, would ask for) -- reported beside `value`, never as `value`
    unordered = None
    if world == 1 and not args.single_stream and not args.force_exchange:
        ctx.set_flags(base_flags | _lib.FLAG_UNORDERED_ROWS)
        for _ in range(args.warmup):
            step()
        barrier()
        t0u = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        unordered = (time.perf_counter() - t0u) / args.steps
        ctx.set_flags(base_flags)
    if args.single_stream:
        timings = ctx.get_timings()
        ctx.set_timing(False)
        kernel_timing_mode = "timed region (one stream)"
    else:
        # The timed region overlaps the event types on separate HIP streams, so a kernel's event-bracketed duration there
        # includes time it shared the GPU with other kernels.  Per-kernel durations (roofline) are therefore taken from a
        # second pass of the same steps with the event types serialised on one stream (== `bench.py --single-stream`, the
        # command the rocprofv3 summary in profiles/ is taken from); every rank takes part.
        ctx.set_flags(base_flags | _lib.FLAG_SINGLE_STREAM)
        for _ in range(args.warmup):
            step()
        barrier()
        ctx.set_timing(True)
        for _ in range(args.steps):
            step()
        barrier()
        timings = ctx.get_timings()
        ctx.set_timing(False)
        ctx.set_flags(base_flags)
        kernel_timing_mode = f"separate single-stream pass of the same steps (the timed region overlaps the event types on {len(shards)} HIP streams)"
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- facts about the last step (identical every step: the build is a pure function of inputs + seed) ----
    inds = [r[0] for r in ctx.results()]
    stats_local = torch.stack([ind.stats for ind in inds]).clone()
    stats = stats_local.clone()
    nnz_out = torch.tensor([int(ind.row_ptr[-1]) for ind in inds], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        dist.all_reduce(nnz_out, op=dist.ReduceOp.SUM)
    stats = stats.cpu().numpy()
    stats_local = stats_local.cpu().numpy()
    nnz_out = nnz_out.cpu().numpy()
    pairs_per_event = [int(s[0]) for s in stats]
    pairs = int(sum(pairs_per_event))
    n_items_a = cfg.events[0].n_items
    items = n_items_a * len(cfg.events)
    ms_per_step = elapsed / args.steps * 1e3
    value = pairs / (elapsed / args.steps)
    nnz_sampled = [int(ind.sampled_row_ptr[-1]) for ind in inds]
    nnz_raw_local = [s.nnz_bound for s in shards]
    item_range = [inds[0].item_lo, inds[0].item_hi]

    if rank != 0:
        ctx.close()
        dist.destroy_process_group()
        return

    # ---- per-kernel table + roofline of the dominant kernel (rank 0's HIP-event timings) ----------------
    facts = []
    NB = _lib.N_BINS
    for d, ev in enumerate(cfg.events):
        st = stats_local[d]     # rank 0's own rows
        facts.append(dict(n_users=hi - lo, nnz_raw=nnz_raw_local[d], nnz_sampled=nnz_sampled[d] // world, nnz_a=nnz_sampled[0], n_items_a=n_items_a,
                          n_items_b=ev.n_items, k=50, bin_rows=[int(x) for x in st[1:1 + NB]], bin_pairs=[int(x) for x in st[1 + NB:1 + 2 * NB]],
                          bin_users=[int(x) for x in st[1 + 2 * NB:1 + 3 * NB]], bin_out=[int(x) for x in st[1 + 3 * NB:1 + 4 * NB]],
                          nnz_out=int(inds[d].row_ptr[-1])))
    per_event_stages = ["column_counts", "downsample_flags", "downsample_scan", "downsample_compact", "row_work"] + BIN_STAGES + ["compact_indicators"]
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
        ms_step = ms / args.steps
        kernels[name] = {"ms_per_step": round(ms_step, 4), "launches_per_step": n // args.steps,
                         "alg_MB_per_step": round(byts / 1e6, 2), "GBps": round(byts / 1e9 / (ms_step / 1e3), 1) if ms_step > 0 and byts > 0 else None}
    scan_ms = sum(kernels[n]["ms_per_step"] for n in ("downsample_flags", "downsample_scan", "downsample_compact") if n in kernels)
    scan_bytes = sum(algorithmic_bytes("csr_row_scan", f) for f in facts)
    kernels["csr_row_scan(flags+scan+compact)"] = {"ms_per_step": round(scan_ms, 4), "alg_MB_per_step": round(scan_bytes / 1e6, 2),
                                                   "GBps": round(scan_bytes / 1e9 / (scan_ms / 1e3), 1) if scan_ms > 0 else None,
                                                   "frac_of_hbm_peak": round(scan_bytes / 1e9 / (scan_ms / 1e3) / HBM_PEAK_GBS, 4) if scan_ms > 0 else None,
                                                   "label": "cache-resident (the matrices fit the 256 MiB Infinity Cache)" if workload == "config3" else "HBM-resident"}
    timed = {k: v for k, v in kernels.items() if not k.startswith("csr_row_scan") and v["GBps"]}
    dominant = max(timed, key=lambda k: timed[k]["ms_per_step"])
    dk = timed[dominant]
    launches = max(dk["launches_per_step"], 1)
    # HBM traffic of the dominant kernel from the committed PMC passes (rocprofv3 cannot wrap the process it runs in)
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "r02_hbm_traffic_pmc.json")
    if world == 1 and workload == "config3" and args.scale == 1.0 and os.path.exists(tpath) and dominant in STAGE_TO_KERNEL:
        tk = json.load(open(tpath))["kernels"].get(STAGE_TO_KERNEL[dominant])
        if tk and "hbm_bytes_per_launch" in tk:
            traffic, traffic_src = tk["hbm_bytes_per_launch"], "profiles/r02_hbm_traffic_pmc.json ((2*FETCH_SIZE + WRITE_SIZE) KB -> bytes, two separate --pmc passes)"
    roofline = {"bound": "hbm", "kernel": dominant, "achieved": dk["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(dk["GBps"] / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "bytes_per_launch": round(dk["alg_MB_per_step"] * 1e6 / launches), "avg_launch_ms": round(dk["ms_per_step"] / launches, 4)}
    # secondary roofline of the SpGEMM classes: LDS accumulate operations (>= one read + one atomic per pair) against the
    # LDS atomic issue peak -- these kernels are neither HBM- nor LDS-throughput bound but dependent-latency bound
    spgemm_ms = sum(kernels[n]["ms_per_step"] for n in BIN_STAGES[:6] if n in kernels)
    spgemm_pairs = sum(sum(f["bin_pairs"][:6]) for f in facts)
    roofline_lds = None
    if spgemm_ms > 0:
        gops = 2.0 * spgemm_pairs / (spgemm_ms / 1e3) / 1e9
        roofline_lds = {"bound": "lds-atomic issue", "kernels": "cco_rows_* (six LDS accumulator classes)", "achieved": round(gops, 1), "peak": round(LDS_ATOMIC_PEAK_GOPS, 1),
                        "unit": "G lane-ops/s", "frac": round(gops / LDS_ATOMIC_PEAK_GOPS, 4), "ops_model": "2 LDS operations per cooccurrence pair (probe read + atomic add)",
                        "note": "far below both ceilings: per-row dependent chains (gather -> insert -> score -> select -> rank) at LDS-limited occupancy"}
    llr_ms = sum(kernels[n]["ms_per_step"] for n in BIN_STAGES + ["compact_indicators", "row_work"] if n in kernels)

    extras = {}
    cpu_baseline = cpu_scipy = None
    if world == 1 and workload == "config3" and args.scale == 1.0:
        ctx.close()
        ctx = None
        if not args.no_extras:
            extras["host_level"] = host_level_leg(library, host_data, cfg, args.seed, pairs)
            extras["csr_row_scan_hbm_resident"] = rowscan_hbm_leg(library, dev, args.seed)
            extras["ingest_to_model"] = ingest_leg(library, dev, host_data, cfg, args.seed)
        if not args.no_cpu_baseline:
            cpu_baseline, cpu_scipy = cpu_legs(host_data, cfg, args.seed, pairs)

    line = {
        "metric": "cooccurrence pairs/sec (A'A+A'B) + LLR top-k items/sec", "value": round(value, 1), "unit": "pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "strong" if world > 1 or workload != "config3" else "weak", "vs_baseline": None, "dtype": "int32 counts / f64 LLR", "data": "synthetic",
        "config": {"workload": names[workload] + ("" if args.scale == 1.0 else f" SCALED x{args.scale} (debug)"),
                   "n_users": cfg.n_users, "n_items": [ev.n_items for ev in cfg.events], "events": [ev.name for ev in cfg.events],
                   "generator": generator, "nnz_raw_rank0": nnz_raw_local, "nnz_sampled": nnz_sampled, "pairs_per_event": pairs_per_event,
                   "maxEventsPerEventType": 500, "maxCorrelatorsPerEventType": 50, "seed": args.seed, "rank0_item_range": item_range,
                   "entry_point": "urcco_context_build_device (C ABI, include/urcco.h)",
                   "parallelism": f"items range-partitioned over {world} GPU(s)" + (", RCCL inside the library: 2 all-reduces + 1 all-gather-v per event type, 1 all-reduce of the work key" if exchange else "")
                                  + ("" if args.single_stream else ", one HIP stream per event type")},
        "pairs_per_step": pairs, "items_per_sec": round(items / (llr_ms / 1e3), 1) if llr_ms > 0 else None,
        "items_per_sec_note": "sum over event types of nItems(A) / time of the SpGEMM+LLR+top-k stages (rank 0)",
        "indicator_entries": int(nnz_out.sum()), "rows_by_accumulator": dict(zip(["micro", "wave", "block_small", "block", "cu_half", "cu", "global"], [int(sum(s[1 + b] for s in stats)) for b in range(NB)])),
        "roofline": roofline, "roofline_lds": roofline_lds, "kernels": kernels, "kernel_timing": kernel_timing_mode, "cpu_baseline": cpu_baseline,
        "cpu_baseline_scipy": cpu_scipy, "gpu_over_cpu": round(value / cpu_baseline["value"], 1) if cpu_baseline else None,
        "input_generation_s": round(gen_s, 1),
        "single_build_latency_ms": None if single_build_ms is None else round(single_build_ms, 4),
        "single_build_latency_note": "median wall time of one build followed by a wait (the timed region enqueues its builds back to back: consecutive builds overlap)",
        "unordered_rows": None if unordered is None else {"flag": "URCCO_FLAG_UNORDERED_ROWS", "ms_per_step": round(unordered * 1e3, 4),
                                                           "pairs_per_s": round(pairs / unordered, 1),
                                                           "note": "same build, indicator rows as unordered top-k sets (no ranking pass); not the headline value"},
    }
    line.update(extras)
    print(json.dumps(line))
    if ctx is not None:
        ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
