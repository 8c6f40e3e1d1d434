#!/usr/bin/env python3
"""Profiling aid: the CSR row scan (urcco_dev_downsample) alone on the config-3 matrices, with parts switched off
(urcco_session_set_debug: 32 = cheap hash, 64 = no threshold gather, 128 = no row lookup).  Outputs of ablated runs
are meaningless.  `--sim` = dry run on the test-only host simulator."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from universal_recommender_amd import _lib, synth  # noqa: E402
from universal_recommender_amd.device import DevCsr, DeviceSession  # noqa: E402

sim = "--sim" in sys.argv
hbm = "--hbm" in sys.argv     # one matrix far beyond the Infinity Cache: config 3's `view` generator with 8M users (what bench.py's csr_row_scan_hbm_resident runs)
c4 = "--config4" in sys.argv  # the five matrices of BASELINE config 4, generated on the device
mode = _lib.RNG_MIX32 if "--rng32" in sys.argv else 0  # the 32-bit form of the down-sampling RNG (URCCO_RNG_MIX32)
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
scale = float(argv[0]) if len(argv) > 0 else 1.0
flags = [int(x) for x in argv[1].split(",")] if len(argv) > 1 else [0, 32, 64, 128, 224]
reps = 2 if sim else 10
dev = torch.device("cpu") if sim else torch.device("cuda", 0)
if hbm or c4:
    cfg = synth.config4(scale) if c4 else synth.config3(1.0)
    if hbm:
        cfg.n_users = int(8_000_000 * scale)
        cfg.events = [cfg.events[1]]
    mats = [DevCsr(cfg.n_users, nc, rp, ci, int(rp[-1].item())) for (_, nc, rp, ci) in synth.generate_device(cfg, dev)]
else:
    cfg = synth.config3(scale)
    data = synth.generate(cfg)
    mats = [DevCsr(cfg.n_users, nc, torch.from_numpy(rp).to(dev), torch.from_numpy(ci).to(dev), int(rp[-1])) for (_, nc, rp, ci) in data]
if sim:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from hostsim import build_sim
    sess = DeviceSession(dev, _lib.load(build_sim.build()))
    sync = lambda: None
else:
    sess = DeviceSession(dev, _lib.load(os.environ.get("URCCO_LIB", _lib.DEFAULT_PATH)))
    sync = torch.cuda.synchronize
raws = [sess.column_counts(m.col_idx, m.nnz_bound, m.n_cols) for m in mats]
for f in flags:
    sess.set_debug(f)
    line = []
    for i, (m, raw) in enumerate(zip(mats, raws)):
        for _ in range(2):
            out, post = sess.downsample(m, m.nnz_bound, raw, 1, 500, mode)
        sync()
        sess.set_timing(True)
        t0 = time.perf_counter()
        for _ in range(reps):
            out, post = sess.downsample(m, m.nnz_bound, raw, 1, 500, mode)
        sync()
        wall = (time.perf_counter() - t0) / reps * 1e3
        tm = sess.get_timings()
        sess.set_timing(False)
        parts = [tm[k][0] / reps for k in ("downsample_flags", "downsample_scan", "downsample_compact")]
        scan = sum(parts)
        kept = int(out.row_ptr[-1].item())
        alg = m.nnz_bound * 4 + kept * 4 + (m.n_rows + 1) * 16
        line.append(f"m{i}: nnz={m.nnz_bound} kept={kept} scan={scan:.4f} ms = " + "+".join(f"{x:.4f}" for x in parts) + f" ({alg / scan / 1e6:.0f} GB/s) wall={wall:.3f}")
    print(f"debug={f}{' rng32' if mode else ''}: " + " | ".join(line), flush=True)
sess.set_debug(0)
