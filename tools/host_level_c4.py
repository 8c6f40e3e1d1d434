#!/usr/bin/env python3
"""The PCIe-inclusive host-level entry point (urcco_cross_occurrence_stage + _finish: what the JNI shim calls) on BASELINE
config 4, outside bench.py: inputs generated on the GPU, mirrored to pageable host memory, then bench.host_level_leg.
URCCO_LIB=other.so for a same-box A/B; URCCO_TRACE_HOST=1 for the library's wall-clock marks of one call."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from universal_recommender_amd import _lib, synth  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "config4"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = {"config3": synth.config3, "config4": synth.config4, "config5": synth.config5}[workload](1.0)
dev = torch.device("cuda", 0)
gen = synth.generate_device(cfg, dev)
host = [(name, nc, rp.cpu().numpy(), ci.cpu().numpy()) for (name, nc, rp, ci) in gen]
del gen
torch.cuda.empty_cache()
lib = _lib.load(os.environ.get("URCCO_LIB", _lib.DEFAULT_PATH))
out = bench.host_level_leg(lib, host, cfg.n_users, 20260925, None, reps=reps)
print(json.dumps({k: out[k] for k in ("ms", "caller_arrays_pinned_ms", "h2d_MB", "d2h_MB", "pairs_per_s")} | {"roofline_pcie": out["roofline_pcie"]["frac"], "lib": os.environ.get("URCCO_LIB", "in-tree")}))
