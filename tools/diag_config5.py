import sys, time, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from helpers import device_generated, check_indicators, to_params
from oracle import c_oracle as O
from universal_recommender_amd import synth, _lib
from universal_recommender_amd import device as D
t0 = time.time()
def mark(s):
    global t0
    torch.cuda.synchronize(); print(f"{time.time()-t0:8.2f}s {s}", flush=True); t0 = time.time()
dev = torch.device("cuda", 0)
cfg = synth.config5(1.0)
dev_mats, mats = device_generated(cfg, dev); mark("generated")
P = lambda: O.DatasetParams(500, 50, None)
params = [P()] * 5
ctx = D.Context(dev, _lib.load(_lib.DEFAULT_PATH), 1, 0, 0)
out = D.cross_occurrence_context(ctx, dev_mats, to_params(params), 20260925); mark("gpu build 1")
out = D.cross_occurrence_context(ctx, dev_mats, to_params(params), 20260925); mark("gpu build 2")
threads = min(os.cpu_count(), O.lib().orc_max_threads())
a = O.downsample(mats[0], O.column_counts(mats[0]), 20260925, 500, 0); mark("oracle downsample A")
cnt_a = O.column_counts(a); a_cp, a_ri = O.transpose(a); mark("oracle transpose")
for d, (m, p, o) in enumerate(zip(mats, params, out)):
    b = a if d == 0 else O.downsample(m, O.column_counts(m), 20260925, 500, 0)
    cnt_b = cnt_a if d == 0 else O.column_counts(b); mark(f"ev{d} oracle downsample")
    ref = O.cco_rows(a_cp, a_ri, b, cnt_a, cnt_b, mats[0].n_rows, d == 0, 50, None, 0, None, threads); mark(f"ev{d} oracle cco_rows")
    got = o.to_host(); mark(f"ev{d} to_host")
    _, ties = check_indicators(got, ref); mark(f"ev{d} check ties={ties}")
