"""Measurement aid: ms per build of config 3 through the context (stream per event type), for one setting of the
URCCO_GRID_FACTORS environment variable (read once per process by the library)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from universal_recommender_amd import _lib, synth, sharded
from universal_recommender_amd.device import DatasetParams, DevCsr
dev = torch.device("cuda", 0)
cfg = synth.config3(1.0)
data = synth.generate(cfg)
shards = [DevCsr(cfg.n_users, nc, torch.from_numpy(rp).to(dev), torch.from_numpy(ci).to(dev), int(rp[-1])) for _, nc, rp, ci in data]
params = [DatasetParams(500, 50, None)] * 3
ctx = sharded.make_context(dev, _lib.load(_lib.DEFAULT_PATH))
out = []
for rep in range(3):
    for _ in range(5):
        ctx.build([[m] for m in shards], params, 1)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        ctx.build([[m] for m in shards], params, 1)
    ctx.synchronize()
    out.append((time.perf_counter() - t0) / 50 * 1e3)
print("factors", os.environ.get("URCCO_GRID_FACTORS", "default"), "ms/step", " ".join(f"{x:.3f}" for x in out), flush=True)
