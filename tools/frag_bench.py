#!/usr/bin/env python3
"""Measurement aid for DESIGN.md section 5: what ONE rank of a W-GPU build spends on getting the primary's CSC of its item range,
on one GPU, for BASELINE config 4's primary (or `--sim`: a dry run on the test-only host simulator at a small scale).

  gathered  (rounds 1-2): urcco_dev_transpose over the WHOLE down-sampled A' restricted to the rank's item range
  fragments (round 3):    urcco_dev_transpose of the rank's own user shard (all columns) + urcco_dev_merge_fragments of the W
                          fragments it receives (the fragments are produced here by transposing every shard in turn)

usage: tools/frag_bench.py [W=8] [--sim] [scale]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from universal_recommender_amd import _lib, synth  # noqa: E402
from universal_recommender_amd.device import DevCsr, DeviceSession  # noqa: E402

sim = "--sim" in sys.argv
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
W = int(argv[0]) if argv else 8
scale = float(argv[1]) if len(argv) > 1 else (0.002 if sim else 1.0)
dev = torch.device("cpu") if sim else torch.device("cuda", 0)
cfg = synth.config4(scale)
cfg.events = cfg.events[:1]
if sim:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from hostsim import build_sim
    sess = DeviceSession(dev, _lib.load(build_sim.build()))
    sync = lambda: None
    (_, nc, rp, ci), = synth.generate(cfg)
    raw_m = DevCsr(cfg.n_users, nc, torch.from_numpy(rp), torch.from_numpy(ci), int(rp[-1]))
else:
    sess = DeviceSession(dev, _lib.load(os.environ.get("URCCO_LIB", _lib.DEFAULT_PATH)))
    sync = torch.cuda.synchronize
    (_, nc, rp, ci), = synth.generate_device(cfg, dev)
    raw_m = DevCsr(cfg.n_users, nc, rp, ci, int(rp[-1].item()))
n_users, n_items = raw_m.n_rows, raw_m.n_cols
raw = sess.column_counts(raw_m.col_idx, raw_m.nnz_bound, n_items)
a, post = sess.downsample(raw_m, raw_m.nnz_bound, raw, 1, 500)
sync()
a_nnz = int(a.row_ptr[-1].item())
a = DevCsr(n_users, n_items, a.row_ptr, a.col_idx, a_nnz)
bounds = sess.partition(sess.row_work_csr(a, a.row_ptr), W)
cuts = [n_users * p // W for p in range(W + 1)]
print(f"A': {n_users} users x {n_items} items, {a_nnz} entries; W = {W}; item bounds {bounds}")


def timed(fn, reps=5):
    for _ in range(2):
        out = fn()
    sync()
    sess.set_timing(True)
    for _ in range(reps):
        out = fn()
    sync()
    tm = sess.get_timings()
    sess.set_timing(False)
    return out, sum(v[0] for v in tm.values()) / reps


shards, frags = [], []
for p in range(W):
    e0, e1 = int(a.row_ptr[cuts[p]].item()), int(a.row_ptr[cuts[p + 1]].item())
    sh = DevCsr(cuts[p + 1] - cuts[p], n_items, (a.row_ptr[cuts[p]:cuts[p + 1] + 1] - e0).contiguous(), a.col_idx[e0:e1].contiguous() if e1 > e0 else a.col_idx[:1].contiguous(), e1 - e0)
    l_cnt = sess.column_counts(sh.col_idx, sh.nnz_bound, n_items)
    (l_cp, l_ri), ms = timed(lambda: sess.transpose(sh, l_cnt))
    shards.append((sh, l_cnt, ms))
    frags.append((l_cnt, l_cp, l_ri))
sizes = torch.zeros(_lib.EXCH_SIZES * W, dtype=torch.int64)
sizes[0::_lib.EXCH_SIZES] = torch.tensor(np.diff(cuts))
sizes = sizes.to(dev)
rows = []
for r in sorted({0, W // 2, W - 1}):
    lo, hi = bounds[r], bounds[r + 1]
    (g_cp, g_ri), ms_gathered = timed(lambda: sess.transpose(a, post, lo, hi))
    lens = torch.cat([f[0][lo:hi] for f in frags])
    assert int(lens.max().item()) <= 0xffff
    lens16 = lens.to(torch.int16).view(torch.uint16) if hasattr(torch, "uint16") else lens
    ents = torch.cat([f[2][int(f[1][lo].item()):int(f[1][hi].item())] for f in frags] + [torch.zeros(1, dtype=torch.int32, device=dev)])
    n_ents = ents.numel() - 1
    (m_cp, m_ri), ms_merge = timed(lambda: sess.merge_fragments(W, lo, hi, n_items, lens16, ents, n_ents, sizes, post))
    assert torch.equal(m_cp, g_cp) and n_ents == int(g_cp[-1].item())
    # same columns, same user sets
    chk = slice(int(g_cp[lo].item()), int(g_cp[lo].item()) + min(n_ents, 200000))
    assert int(m_ri[:n_ents].to(torch.int64).sum().item()) == int(g_ri[:n_ents].to(torch.int64).sum().item())
    rows.append((r, hi - lo, n_ents, ms_gathered, shards[r][2], ms_merge))
    print(f"rank {r}: range [{lo}, {hi}) = {hi - lo} items, {n_ents} entries | gathered pass {ms_gathered:.3f} ms | fragments: own-shard transposition "
          f"{shards[r][2]:.3f} + merge {ms_merge:.3f} = {shards[r][2] + ms_merge:.3f} ms | wire: {2 * (hi - lo) * W + 4 * n_ents} B in, vs {4 * a_nnz} B of A' entries gathered either way")
