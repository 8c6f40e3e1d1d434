#!/usr/bin/env python3
"""Probe (round 6, VERDICT r05 #1b): what would a locality-aware USER ORDER buy the SpGEMM classes?  Config 4 is built twice on one box: as generated,
and with the users renumbered so that users whose primary row ends in the same item are neighbours (the users of a rare item -- most item rows are rare
items -- then own consecutive B' rows: the row walk of that item reads one contiguous stretch instead of scattered ~100-byte rows).  The renumbering is
done here, outside the timed builds (torch sort + gathers); the down-sampling RNG is keyed by the user id, so the two builds sample different entries of
the same distribution -- per-class times and pair counts are compared, not rows.  Not a product path."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from universal_recommender_amd import _lib, synth  # noqa: E402
from universal_recommender_amd.device import DatasetParams, DevCsr, DeviceSession, cross_occurrence_device  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
key_kind = sys.argv[2] if len(sys.argv) > 2 else "last"   # last (largest item id of the row) | first | random (control: a random renumbering)
dev = torch.device("cuda", 0)
cfg = synth.config4(scale)
raw = [(nc, rp, ci) for (_, nc, rp, ci) in synth.generate_device(cfg, dev)]
n_users = cfg.n_users


def renumber(perm):
    out = []
    for nc, rp, ci in raw:
        lens = rp[1:] - rp[:-1]
        nl = lens[perm]
        nrp = torch.zeros(n_users + 1, dtype=torch.int64, device=dev)
        torch.cumsum(nl, 0, out=nrp[1:])
        src0 = rp[:-1][perm]
        idx = torch.repeat_interleave(src0 - nrp[:-1], nl) + torch.arange(int(nrp[-1].item()), device=dev)
        out.append((nc, nrp, ci[idx].contiguous()))
        del idx
    return out


nc0, rp0, ci0 = raw[0]
lens0 = rp0[1:] - rp0[:-1]
if key_kind == "random":
    perm = torch.randperm(n_users, device=dev)
else:
    pos = (rp0[1:] - 1) if key_kind == "last" else rp0[:-1]
    key = torch.where(lens0 > 0, ci0[pos.clamp(min=0, max=ci0.numel() - 1)].to(torch.int64), torch.full_like(lens0, 1 << 40))
    perm = torch.argsort(key, stable=True)
variants = {"as generated": raw, f"users renumbered by their {key_kind} primary item": renumber(perm)}
params = [DatasetParams(500, 50, None) for _ in raw]
sess = DeviceSession(dev, _lib.load(os.environ.get("URCCO_LIB", _lib.DEFAULT_PATH)))
for name, mats_ in variants.items():
    mats = [DevCsr(n_users, nc, rp, ci, int(rp[-1].item())) for (nc, rp, ci) in mats_]
    for _ in range(2):
        out = cross_occurrence_device(sess, mats, params, 1)
    torch.cuda.synchronize()
    pairs = sum(int(o.stats[0].item()) for o in out)
    sess.set_timing(True)
    steps = 5
    t0 = time.perf_counter()
    for _ in range(steps):
        cross_occurrence_device(sess, mats, params, 1)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    tm = sess.get_timings()
    sess.set_timing(False)
    rows = sum(v[0] for k, v in tm.items() if k.startswith("cco_rows")) / steps
    print(f"{name}: pairs {pairs} | serialised session {wall:.2f} ms/build | SpGEMM classes {rows:.3f} ms | " +
          " ".join(f"{k.replace('cco_rows_', '')}={v[0] / steps:.3f}" for k, v in tm.items() if v[1] and (k.startswith('cco_rows') or k in ('row_work', 'transpose', 'entropy'))), flush=True)
sess.close()
