#!/usr/bin/env python3
"""Profiling aid: numNonZeroElementsPerColumn (urcco_dev_column_counts) alone on the raw matrices of BASELINE config 4 (generated on the
device), the part-local layout of round 5 against the bucket-contiguous one (URCCO_COLCOUNT_GLOBAL_LAYOUT=1), counts compared with
torch.bincount.  usage: tools/colcount_bench.py [scale] [--only-new]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from universal_recommender_amd import _lib, synth  # noqa: E402
from universal_recommender_amd.device import DevCsr, DeviceSession  # noqa: E402

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
scale = float(argv[0]) if argv else 1.0
dev = torch.device("cuda", 0)
cfg = synth.config4(scale)
mats = [DevCsr(cfg.n_users, nc, rp, ci, int(rp[-1].item())) for (_, nc, rp, ci) in synth.generate_device(cfg, dev)]
sess = DeviceSession(dev, _lib.load(os.environ.get("URCCO_LIB", _lib.DEFAULT_PATH)))
reps = 10
layouts = [("part-local", "0")] if "--only-new" in sys.argv else [("part-local", "0"), ("bucket-contiguous", "1")]
for name, v in layouts:
    os.environ["URCCO_COLCOUNT_GLOBAL_LAYOUT"] = v
    tot = 0.0
    line = []
    for i, m in enumerate(mats):
        cnt = sess.column_counts(m.col_idx, m.nnz_bound, m.n_cols)
        torch.cuda.synchronize()
        ok = bool(torch.equal(cnt[:m.n_cols].to(torch.int64), torch.bincount(m.col_idx[:m.nnz_bound].to(torch.int64), minlength=m.n_cols)))
        sess.set_timing(True)
        for _ in range(reps):
            sess.column_counts(m.col_idx, m.nnz_bound, m.n_cols)
        torch.cuda.synchronize()
        tm = sess.get_timings()
        sess.set_timing(False)
        ms = tm["column_counts"][0] / reps
        tot += ms
        line.append(f"m{i}: nnz={m.nnz_bound} cols={m.n_cols} {ms:.4f} ms ({4 * m.nnz_bound / ms / 1e6:.0f} GB/s alg) {'ok' if ok else 'WRONG'}")
    print(f"{name}: total {tot:.3f} ms | " + " | ".join(line), flush=True)
sess.close()
