#!/usr/bin/env python3
"""Profiling aid: where a sharded (multi-GPU path) model build spends its time on ONE GPU (one-rank nccl group,
force_exchange): host clock and GPU event clock at the phase boundaries of sharded.cross_occurrence_sharded.
The collectives move no data between devices here; what shows is the host-side cost of the path (the one host sync,
enqueue rate) against the single-GPU driver."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)

from universal_recommender_amd import _lib, sharded, synth  # noqa: E402
from universal_recommender_amd.device import DatasetParams, DevCsr, DeviceSession, SessionPool  # noqa: E402

cfg = synth.config3(float(sys.argv[1]) if len(sys.argv) > 1 else 1.0)
data = synth.generate(cfg)
mats = [DevCsr(cfg.n_users, nc, torch.from_numpy(rp).to(dev), torch.from_numpy(ci).to(dev), int(rp[-1])) for (_, nc, rp, ci) in data]
params = [DatasetParams(500, 50, None) for _ in mats]
lib = _lib.load(_lib.DEFAULT_PATH)
sess = DeviceSession(dev, lib)
pool = SessionPool(dev, len(mats), lib)

marks = []
orig = {}


def wrap(obj, name, label):
    f = getattr(obj, name)
    orig[(obj, name)] = f

    def g(*a, **k):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream(dev))
        marks.append((label + ":in", time.perf_counter(), ev))
        r = f(*a, **k)
        ev2 = torch.cuda.Event(enable_timing=True)
        ev2.record(torch.cuda.current_stream(dev))
        marks.append((label + ":out", time.perf_counter(), ev2))
        return r
    setattr(obj, name, g)


for owner in (sess, pool[0]):
    wrap(owner, "partition", "partition(host sync)")
    wrap(owner, "transpose", "transpose")
wrap(sharded, "_gather_start", "gather_start")
wrap(sharded, "_gather_finish", "gather_finish")
wrap(sharded, "_exchange_sizes_finish", "sizes(host read)")
for use_pool in (False, True):
    for it in range(4):
        marks.clear()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(dev))
        t0 = time.perf_counter()
        res = sharded.cross_occurrence_sharded(sess, mats, params, 1, cfg.n_users, 0, force_exchange=True, pool=pool if use_pool else None)
        t_enq = time.perf_counter()
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record(torch.cuda.current_stream(dev))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
    print(f"pool={use_pool}: step wall {1e3 * (t1 - t0):.3f} ms, host done enqueueing at {1e3 * (t_enq - t0):.3f} ms, GPU e0->e1 {e0.elapsed_time(e1):.3f} ms")
    for label, th, ev in marks:
        print(f"   {label:28s} host {1e3 * (th - t0):7.3f} ms   gpu {e0.elapsed_time(ev):7.3f} ms")
dist.destroy_process_group()
