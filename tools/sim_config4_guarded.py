#!/usr/bin/env python3
"""Debug aid: BASELINE config 4 at a small scale through urcco_context_build_device on the TEST-ONLY host simulator, three builds back to back;
with HIPSIM_GUARD=1|2 every buffer ends at a guard page and fresh / re-issued scratch memory is poisoned (tests/hostsim/hipsim.cpp).
usage: HIPSIM_GUARD=1 [SIM_CONFIG5=1] tools/sim_config4_guarded.py SCALE FLAGS [ITEM_SCALE]   (FLAGS 1 = single stream)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
from hostsim import build_sim
from universal_recommender_amd import _lib, synth
from universal_recommender_amd.device import Context, DatasetParams, DevCsr
import conftest, helpers
lib = _lib.load(build_sim.build())
scale = float(sys.argv[1]); flags = int(sys.argv[2])
item_scale = float(sys.argv[3]) if len(sys.argv) > 3 else None   # 1.0 keeps the 2M-wide item spaces (245 column buckets, 21-bit keys)
cfg = (synth.config5 if os.environ.get("SIM_CONFIG5") else synth.config4)(scale, item_scale)
data = synth.generate(cfg)
helpers.GUARD_LIB = lib if os.environ.get("HIPSIM_GUARD") else None
mats = []
for (_, nc, rp, ci) in data:
    mats.append([DevCsr(cfg.n_users, nc, helpers.guarded(torch.from_numpy(rp)), helpers.guarded(torch.from_numpy(ci if ci.size else np.zeros(1, np.int32))), int(rp[-1]))])
print("nnz", [m[0].nnz_bound for m in mats], flush=True)
ctx = Context(torch.device("cpu"), lib, 1, flags)
params = [DatasetParams(500, 50, None)] * len(mats)
if os.environ.get("SIM_TIMING"):      # HIP-event stage timing on: candidate counters, per-bin output statistics
    ctx.set_timing(True)
t0 = time.time()
for i in range(3):
    ctx.build(mats, params, 20260925, cfg.n_users, [0])
ctx.synchronize()
res = ctx.results()
print("pairs", [int(r[0].stats[0]) for r in res], "time", round(time.time() - t0, 1))
ctx.close()
