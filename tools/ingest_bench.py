#!/usr/bin/env python3
"""Profiling aid: the device-side Preparator (ingest kernels) on config-3-shaped event streams: per event type the
(user, item) interactions of the synthetic generator in random stream order with ~10 % duplicate events, as 64-bit keys
resident in HBM.  Prints wall time per phase (dictionary builds, lookups, CSR build) and events/s.  `--sim` = dry run on
the test-only host simulator at a small scale.  Not part of bench.py's metric."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from universal_recommender_amd import _lib, ingest, synth  # noqa: E402
from universal_recommender_amd.device import DeviceSession  # noqa: E402

sim = "--sim" in sys.argv
argv = [a for a in sys.argv[1:] if a != "--sim"]
scale = float(argv[0]) if argv else (0.01 if sim else 1.0)
dev = torch.device("cpu") if sim else torch.device("cuda", 0)
if sim:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from hostsim import build_sim
    sess = DeviceSession(dev, _lib.load(build_sim.build()))
    sync = lambda: None
else:
    sess = DeviceSession(dev, _lib.load(_lib.DEFAULT_PATH))
    sync = torch.cuda.synchronize

cfg = synth.config3(scale)
rng = np.random.default_rng(1)
actions = []
total = 0
for (name, n_cols, rp, ci) in synth.generate(cfg):
    rows = np.repeat(np.arange(cfg.n_users, dtype=np.int64), np.diff(rp))
    cols = ci.astype(np.int64)
    dup = rng.integers(0, rows.size, rows.size // 10)                      # ~10 % repeated events
    rows, cols = np.concatenate([rows, rows[dup]]), np.concatenate([cols, cols[dup]])
    order = rng.permutation(rows.size)
    uk = (rows[order] * np.int64(0x9E3779B97F4A7C15 - (1 << 64)) + 11)     # any injective map to 64-bit keys
    ik = (cols[order] * np.int64(0xC2B2AE3D27D4EB4F - (1 << 64)) + 5)
    actions.append((name, torch.from_numpy(uk).to(dev), torch.from_numpy(ik).to(dev)))
    total += rows.size
for it in range(3):
    sync()
    t0 = time.perf_counter()
    got = ingest.prepare_device(sess, actions, 1)
    sess.synchronize()
    sync()
    dt = time.perf_counter() - t0
    print(f"run {it}: {total} events -> users {got.user_first_pos.numel()}, nnz {[e.matrix.nnz_bound for e in got.events]} in {dt * 1e3:.2f} ms "
          f"= {total / dt / 1e6:.1f} M events/s (includes 2 dictionary builds + 2 lookups + CSR build per event type, host syncs, allocations)")
