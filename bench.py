#!/usr/bin/env python3
"""bench.py -- CCO model-build throughput on MI355X (BASELINE.json metric: cooccurrence pairs/sec (A'A + A'B) + LLR
top-k items/sec).

A "step" = one complete pass of the hot path over the workload with the raw per-event user x item CSR matrices
already resident in HBM: column counts -> sampleDownAndBinarize -> A.t -> per event type A.t %*% B fused with LLR +
top-k (+ the RCCL all-reduces / all-gather for N > 1).  Nothing is cached between steps.

  N = 1   workload = BASELINE config 3 (synthetic 1M users x 200K items, Zipf-1.0, purchase/view/category-pref) --
          configs[1] (the 30-user MovieLens sample, 35K pairs) is a parity-test case, it cannot load a GPU.
  N > 1   weak scaling: config 3 grown N-fold in users AND items (N = 8 -> 8M x 1.6M, config-4 scale); rank r
          generates and owns users [r, r+1) * 1M; items are range-partitioned by work (sharded.py).

value = cooccurrence pairs formed per second, whole job (all ranks), max-over-ranks time.  One JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes(stage: str, f: dict) -> float:
    """Algorithmic HBM bytes of one launch of a stage (int32 indices, implicit values; L2-resident gathers
    of per-column counts are not counted).  f = per-event-type facts."""
    U, nnz, nnzs, IA, IB, k = f["n_users"], f["nnz_raw"], f["nnz_sampled"], f["n_items_a"], f["n_items_b"], f["k"]
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
        b = {"cco_rows_micro": 0, "cco_rows_wave": 1, "cco_rows_block_small": 2, "cco_rows_block": 3, "cco_rows_cu_half": 4, "cco_rows_cu": 5, "cco_rows_global": 6}[stage]
        rows, pairs, users, outs = f["bin_rows"][b], f["bin_pairs"][b], f["bin_users"][b], f["bin_out"][b]
        return 4.0 * rows + 16.0 * rows + 4.0 * users + 16.0 * users + 4.0 * pairs + 12.0 * outs + 4.0 * rows
    if stage == "compact_indicators":
        return 16.0 * IA * 1.0 + 24.0 * f["nnz_out"]
    return 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the workload (debug only; the reported config says so)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--single-stream", action="store_true", help="N = 1: run the event types back to back on one HIP stream")
    ap.add_argument("--force-exchange", action="store_true",
                    help="debug: run the N > 1 code path (RCCL collectives, range-restricted transpose) in a one-rank group")
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
    distributed = world > 1 or args.force_exchange
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    from universal_recommender_amd import _lib, sharded, synth
    from universal_recommender_amd.device import DatasetParams, DevCsr, DeviceSession
    if not os.path.exists(_lib.DEFAULT_PATH):   # the in-tree HIP library normally travels with the repo; build it otherwise
        if local_rank == 0:
            import __graft_entry__
            __graft_entry__.build_hip()
        if distributed:
            dist.barrier()

    # ---- workload -------------------------------------------------------------------------------------
    base = synth.config3(args.scale)
    users_per_rank = base.n_users
    cfg = synth.config3(args.scale)
    if world > 1:  # weak scaling: N x users and N x items
        cfg.n_users = users_per_rank * world
        for ev in cfg.events:
            if ev.n_items > 5000:
                ev.n_items *= world
        cfg.name = f"config3 x{world} (weak scaling: {cfg.n_users} users x {cfg.events[0].n_items} items, 3 events)"
    lo, hi = rank * users_per_rank, (rank + 1) * users_per_rank
    t0 = time.time()
    data = synth.generate(cfg, lo, hi)
    gen_s = time.time() - t0
    params = [DatasetParams(500, 50, None) for _ in data]   # engine.json defaults: maxEventsPerEventType 500, maxCorrelatorsPerEventType 50
    shards = [DevCsr(hi - lo, nc, torch.from_numpy(rp).to(dev), torch.from_numpy(ci).to(dev), int(rp[-1])) for (_, nc, rp, ci) in data]
    from universal_recommender_amd.device import SessionPool, cross_occurrence_streams
    library = _lib.load(_lib.DEFAULT_PATH)
    pool = None
    if distributed:
        sess = DeviceSession(dev, library)
        if not args.single_stream:
            pool = SessionPool(dev, len(shards), library)   # the A'B_d of each event type on its own HIP stream
    else:
        sess = SessionPool(dev, 1 if args.single_stream else len(shards), library)   # one HIP stream per event type

    def step(use_pool=True):
        if distributed:
            return sharded.cross_occurrence_sharded(sess, shards, params, args.seed, cfg.n_users, lo, force_exchange=args.force_exchange,
                                                    pool=pool if use_pool else None)
        return sharded.ShardedResult(cross_occurrence_streams(sess, shards, params, args.seed), [[0, shards[0].n_cols]] * len(shards), [-1] * len(shards))

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        res = step()
    barrier()
    serial_pass = not args.single_stream and (pool is not None or (not distributed and len(sess) > 1))
    if not serial_pass:
        sess.set_timing(True)   # HIP events around every launch group, on the launching stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    barrier()
    elapsed = time.perf_counter() - t0
    timings = sess.get_timings() if not serial_pass else {}
    sess.set_timing(False)
    kernel_timing_mode = "timed region (one stream)"
    if serial_pass and distributed:
        # per-kernel durations from a second pass with every launch on the one main stream (all ranks take part)
        for _ in range(args.warmup):
            step(use_pool=False)
        barrier()
        sess.set_timing(True)
        for _ in range(args.steps):
            res = step(use_pool=False)
        barrier()
        timings = sess.get_timings()
        sess.set_timing(False)
        kernel_timing_mode = "separate single-stream pass of the same steps (the timed region runs the event types on separate HIP streams)"
    elif serial_pass:
        # The timed region overlaps the event types on separate HIP streams, so a kernel's event-bracketed duration there
        # includes time it shared the GPU with other kernels.  Per-kernel durations (roofline) are therefore taken from a
        # second pass of the same steps with the event types serialised on one stream (== `bench.py --single-stream`,
        # the command the rocprofv3 summary in profiles/ is taken from).
        from universal_recommender_amd.device import cross_occurrence_device
        one = sess[0]
        with torch.cuda.stream(one.torch_stream):
            for _ in range(args.warmup):
                cross_occurrence_device(one, shards, params, args.seed)
            torch.cuda.synchronize(dev)
            one.set_timing(True)
            for _ in range(args.steps):
                res1 = cross_occurrence_device(one, shards, params, args.seed)
            torch.cuda.synchronize(dev)
            timings = one.get_timings()
            one.set_timing(False)
        res = sharded.ShardedResult(res1, res.item_ranges, res.nnz_sampled)   # carries the per-bin emitted-entry stats
        kernel_timing_mode = "separate single-stream pass of the same steps (the timed region overlaps event types on 3 HIP streams)"
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- facts about the last step (identical every step: the build is a pure function of inputs + seed) ----
    stats = torch.stack([ind.stats for ind in res.indicators]).clone()
    nnz_out = torch.tensor([int(ind.row_ptr[-1]) for ind in res.indicators], dtype=torch.int64, device=dev)
    if distributed:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        dist.all_reduce(nnz_out, op=dist.ReduceOp.SUM)
    stats = stats.cpu().numpy()
    nnz_out = nnz_out.cpu().numpy()
    pairs_per_event = [int(s[0]) for s in stats]
    pairs = int(sum(pairs_per_event))
    n_items_a = cfg.events[0].n_items
    items = n_items_a * len(cfg.events)
    ms_per_step = elapsed / args.steps * 1e3
    value = pairs / (elapsed / args.steps)

    if rank != 0:
        dist.destroy_process_group()
        return

    # ---- per-kernel table + roofline of the dominant kernel (rank 0's HIP-event timings) ----------------
    nnz_sampled = [int(ind.sampled_row_ptr[-1]) for ind in res.indicators]
    facts = []
    NB = _lib.N_BINS
    for d, ev in enumerate(cfg.events):
        st = stats[d] if world == 1 else torch.stack([ind.stats for ind in res.indicators])[d].cpu().numpy()  # rank 0's own rows
        facts.append(dict(n_users=cfg.n_users, nnz_raw=shards[d].nnz_bound, nnz_sampled=nnz_sampled[d] if world > 1 else nnz_sampled[d],
                          nnz_a=nnz_sampled[0], n_items_a=n_items_a, n_items_b=ev.n_items, k=50,
                          bin_rows=[int(x) for x in st[1:1 + NB]], bin_pairs=[int(x) for x in st[1 + NB:1 + 2 * NB]],
                          bin_users=[int(x) for x in st[1 + 2 * NB:1 + 3 * NB]], bin_out=[int(x) for x in st[1 + 3 * NB:1 + 4 * NB]],
                          nnz_out=int(res.indicators[d].row_ptr[-1])))
    per_event_stages = ["column_counts", "downsample_flags", "downsample_scan", "downsample_compact", "row_work", "cco_rows_micro",
                        "cco_rows_wave", "cco_rows_block_small", "cco_rows_block", "cco_rows_cu_half", "cco_rows_cu", "cco_rows_global", "compact_indicators"]
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
                                                   "frac_of_hbm_peak": round(scan_bytes / 1e9 / (scan_ms / 1e3) / HBM_PEAK_GBS, 4) if scan_ms > 0 else None}
    timed = {k: v for k, v in kernels.items() if not k.startswith("csr_row_scan") and v["GBps"]}
    dominant = max(timed, key=lambda k: timed[k]["ms_per_step"])
    dk = timed[dominant]
    launches = max(dk["launches_per_step"], 1)
    # HBM traffic of the dominant kernel from the committed PMC passes (rocprofv3 cannot wrap the process it runs in)
    traffic, traffic_src = None, None
    stage_to_kernel = {"cco_rows_micro": "cco_rows_micro_kernel", "cco_rows_wave": "cco_rows_kernel<64, 1024>",
                       "cco_rows_block_small": "cco_rows_kernel<256, 4096>", "cco_rows_block": "cco_rows_kernel<256, 8192>",
                       "cco_rows_cu_half": "cco_rows_kernel<512, 16384>", "cco_rows_cu": "cco_rows_kernel<1024, 32768>", "downsample_flags": "downsample_flags_kernel",
                       "transpose": "transpose_kernel"}
    tpath = os.path.join(ROOT, "profiles", "r01_hbm_traffic_pmc.json")
    if world == 1 and args.scale == 1.0 and os.path.exists(tpath) and dominant in stage_to_kernel:
        tk = json.load(open(tpath))["kernels"].get(stage_to_kernel[dominant])
        if tk:
            traffic, traffic_src = tk["hbm_bytes_per_launch"], "profiles/r01_hbm_traffic_pmc.json (2*FETCH_SIZE + WRITE_SIZE, KB -> bytes)"
    roofline = {"bound": "hbm", "kernel": dominant, "achieved": dk["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(dk["GBps"] / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "bytes_per_launch": round(dk["alg_MB_per_step"] * 1e6 / launches), "avg_launch_ms": round(dk["ms_per_step"] / launches, 4)}
    llr_ms = sum(kernels[n]["ms_per_step"] for n in ("cco_rows_micro", "cco_rows_wave", "cco_rows_block_small", "cco_rows_block", "cco_rows_cu_half", "cco_rows_cu", "cco_rows_global",
                                                      "compact_indicators", "row_work") if n in kernels)

    # ---- CPU baseline: the C oracle (a restatement of the Mahout algorithm -- Mahout/Spark itself cannot run here:
    #      no JVM, un-vendored jars) on the same workload, host cores of this box -------------------------
    cpu_baseline = None
    if world == 1 and not args.no_cpu_baseline:
        from oracle import c_oracle as O
        cores = min(os.cpu_count() or 1, O.lib().orc_max_threads())
        mats = [O.Csr(cfg.n_users, nc, rp, ci) for (_, nc, rp, ci) in data]
        t0 = time.perf_counter()
        ref = O.cross_occurrence_downsampled(mats, [O.DatasetParams(500, 50, None)] * len(mats), args.seed, 0, cores)
        cpu_s = time.perf_counter() - t0
        cpu_pairs = sum(r.pairs for r in ref)
        cpu_baseline = {"value": round(cpu_pairs / cpu_s, 1), "unit": "pairs/s", "cores": cores, "kind": "port",
                        "sample": f"the whole workload once ({cpu_pairs} pairs in {cpu_s:.2f} s); C oracle, OpenMP over item rows, "
                                  "down-sampling/transpose single-threaded; Mahout/Spark local[*] is not runnable in this image",
                        "pairs_match_gpu": bool(cpu_pairs == pairs)}

    line = {
        "metric": "cooccurrence pairs/sec (A'A+A'B) + LLR top-k items/sec", "value": round(value, 1), "unit": "pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "int32 counts / f64 LLR", "data": "synthetic",
        "config": {"workload": cfg.name if world > 1 else ("config3: synthetic 1M users x 200K items, Zipf-1.0, purchase/view/category-pref"
                                                             + ("" if args.scale == 1.0 else f" SCALED x{args.scale} (debug)")),
                   "n_users": cfg.n_users, "n_items": [ev.n_items for ev in cfg.events], "events": [ev.name for ev in cfg.events],
                   "nnz_raw": [s.nnz_bound for s in shards] if world == 1 else None, "nnz_sampled": nnz_sampled,
                   "pairs_per_event": pairs_per_event, "maxEventsPerEventType": 500, "maxCorrelatorsPerEventType": 50, "seed": args.seed,
                   "parallelism": f"items range-partitioned over {world} GPU(s)" + (", RCCL: 3 all-reduces + asynchronous all-gathers per event type" if world > 1 else "")
                                  + ("" if args.single_stream else ", one HIP stream per event type")},
        "pairs_per_step": pairs, "items_per_sec": round(items / (llr_ms / 1e3), 1) if llr_ms > 0 else None,
        "items_per_sec_note": "sum over event types of nItems(A) / time of the SpGEMM+LLR+top-k stages",
        "indicator_entries": int(nnz_out.sum()), "rows_by_accumulator": dict(zip(["micro", "wave", "block_small", "block", "cu_half", "cu", "global"], [int(sum(s[1 + b] for s in stats)) for b in range(_lib.N_BINS)])),
        "roofline": roofline, "kernels": kernels, "kernel_timing": kernel_timing_mode, "cpu_baseline": cpu_baseline,
        "gpu_over_cpu": round(value / cpu_baseline["value"], 1) if cpu_baseline else None,
        "host_generation_s": round(gen_s, 1),
    }
    print(json.dumps(line))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
