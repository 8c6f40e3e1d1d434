#!/usr/bin/env python3
"""Profiling aid: per-stage HIP-event timings of the config-3 model build with kernel phases switched off
(urcco_session_set_debug: 1 = gather only, 2 = no LLR, 4 = no top-k, 8 = no select, 16 = no rank).  Results of ablated runs are meaningless."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from universal_recommender_amd import _lib, synth  # noqa: E402
from universal_recommender_amd.device import DatasetParams, DevCsr, DeviceSession, cross_occurrence_device  # noqa: E402

c4 = "--config4" in sys.argv      # BASELINE config 4 (HBM-resident), generated on the device
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
scale = float(argv[0]) if len(argv) > 0 else 1.0
flags = [int(x) for x in argv[1].split(",")] if len(argv) > 1 else [0, 4, 6, 7]
dev = torch.device("cuda", 0)
if c4:
    cfg = synth.config4(scale)
    mats = [DevCsr(cfg.n_users, nc, rp, ci, int(rp[-1].item())) for (_, nc, rp, ci) in synth.generate_device(cfg, dev)]
else:
    cfg = synth.config3(scale)
    data = synth.generate(cfg)
    mats = [DevCsr(cfg.n_users, nc, torch.from_numpy(rp).to(dev), torch.from_numpy(ci).to(dev), int(rp[-1])) for (_, nc, rp, ci) in data]
params = [DatasetParams(500, 50, None) for _ in mats]
sess = DeviceSession(dev, _lib.load(os.environ.get("URCCO_LIB", _lib.DEFAULT_PATH)))
for f in flags:
    sess.set_debug(f)
    for _ in range(2):
        cross_occurrence_device(sess, mats, params, 1)
    torch.cuda.synchronize()
    sess.set_timing(True)
    t0 = time.perf_counter()
    steps = 3
    for _ in range(steps):
        cross_occurrence_device(sess, mats, params, 1)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    tm = sess.get_timings()
    sess.set_timing(False)
    print(f"debug={f} wall {wall:.3f} ms/step | " + " ".join(f"{k.replace('cco_rows_', '')}={v[0] / steps:.3f}" for k, v in tm.items() if v[1] and k.startswith("cco_rows")))
sess.set_debug(0)
