#!/usr/bin/env python3
"""Profiling aid: the CSR row scan (urcco_dev_downsample) alone on the config-3 matrices, with parts switched off
(urcco_session_set_debug: 256 = three-pass form, 32 = cheap hash, 64 = no threshold gather, 128 = no row lookup).
Outputs of ablated runs are meaningless; only the unablated forms are compared with each other."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from universal_recommender_amd import _lib, synth  # noqa: E402
from universal_recommender_amd.device import DevCsr, DeviceSession  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
flags = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 256, 32, 64, 128, 224]
reps = 10
dev = torch.device("cuda", 0)
cfg = synth.config3(scale)
data = synth.generate(cfg)
mats = [DevCsr(cfg.n_users, nc, torch.from_numpy(rp).to(dev), torch.from_numpy(ci).to(dev), int(rp[-1])) for (_, nc, rp, ci) in data]
sess = DeviceSession(dev, _lib.load(_lib.DEFAULT_PATH))
raws = [sess.column_counts(m.col_idx, m.nnz, m.n_cols) for m in mats]
outs = {}
for f in flags:
    sess.set_debug(f)
    line = []
    for i, (m, raw) in enumerate(zip(mats, raws)):
        for _ in range(2):
            out, post = sess.downsample(m, m.nnz, raw, 1, 500)
        torch.cuda.synchronize()
        sess.set_timing(True)
        t0 = time.perf_counter()
        for _ in range(reps):
            out, post = sess.downsample(m, m.nnz, raw, 1, 500)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps * 1e3
        tm = sess.get_timings()
        sess.set_timing(False)
        scan = sum(tm[k][0] for k in ("downsample_flags", "downsample_scan", "downsample_compact")) / reps
        kept = int(out.row_ptr[-1].item())
        alg = m.nnz * 4 + kept * 4 + (m.n_rows + 1) * 16
        line.append(f"m{i}: nnz={m.nnz} kept={kept} scan={scan:.4f} ms ({alg / scan / 1e6:.0f} GB/s) wall={wall:.3f}")
        if f in (0, 256):
            outs[(f, i)] = (out.row_ptr.clone(), out.col_idx[:kept].clone(), post.clone())
    print(f"debug={f}: " + " | ".join(line), flush=True)
sess.set_debug(0)
if 0 in flags and 256 in flags:
    same = all(torch.equal(a, b) for i in range(len(mats)) for a, b in zip(outs[(0, i)], outs[(256, i)]))
    print("single-pass == three-pass:", same)
