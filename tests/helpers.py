"""Shared test helpers: seeded matrix generators, host<->device conversion, the tie-aware parity checker."""
from __future__ import annotations

import os

import numpy as np
import torch

from oracle import c_oracle as O
from universal_recommender_amd import device as D

LLR_TOL = 1e-6  # BASELINE.json north_star: "same ids, LLR scores within 1e-6"


def rand_csr(rng, n_rows, n_cols, avg, zipf_s=1.0, empty_frac=0.0):
    """Binary CSR with Poisson(avg) entries per row drawn from a bounded Zipf(zipf_s) over a permuted id space."""
    w = 1.0 / np.power(np.arange(1, n_cols + 1, dtype=np.float64), zipf_s)
    cdf = np.cumsum(w) / w.sum()
    perm = rng.permutation(n_cols)
    deg = rng.poisson(avg, n_rows)
    if empty_frac > 0:
        deg[rng.random(n_rows) < empty_frac] = 0
    total = int(deg.sum())
    ranks = np.minimum(np.searchsorted(cdf, rng.random(total)), n_cols - 1)
    users = np.repeat(np.arange(n_rows, dtype=np.int64), deg)
    key = np.unique(users * n_cols + perm[ranks])
    u = key // n_cols
    rp = np.zeros(n_rows + 1, np.int64)
    np.cumsum(np.bincount(u, minlength=n_rows), out=rp[1:])
    return O.Csr(n_rows, n_cols, rp, (key - u * n_cols).astype(np.int32))


GUARD_LIB = None  # set by conftest.sim_session under HIPSIM_GUARD: inputs of simulator runs end at guard pages


def guarded(t: torch.Tensor) -> torch.Tensor:
    if GUARD_LIB is None or t.device.type != "cpu":
        return t
    from conftest import guarded_tensor
    g = guarded_tensor(GUARD_LIB, t.numel(), t.dtype)
    g.copy_(t.reshape(-1))
    return g


def to_dev(m: O.Csr, device) -> D.DevCsr:
    ci = m.col_idx if m.nnz else np.zeros(1, np.int32)
    if GUARD_LIB is not None and torch.device(device).type == "cpu":
        return D.DevCsr(m.n_rows, m.n_cols, guarded(torch.from_numpy(m.row_ptr.copy())), guarded(torch.from_numpy(ci[: max(m.nnz, 1)].copy())), m.nnz)
    return D.DevCsr(m.n_rows, m.n_cols, torch.from_numpy(m.row_ptr.copy()).to(device), torch.from_numpy(ci.copy()).to(device), m.nnz)


def to_params(ps):
    return [D.DatasetParams(p.max_elements_per_row, p.max_interesting_elements, p.min_llr) for p in ps]


def check_indicators(got, ref: O.IndicatorRows, tol=LLR_TOL, exact_ids=False):
    """got = (row_ptr, col_idx, llr) CSR from the HIP path; ref = oracle rows.
    Parity bar: bit-exact ids (tie-aware at the k-th score when LLRs differ in the last ulps), |dLLR| <= tol.
    Returns (n_rows_checked, n_rows_needing_the_tie_rule)."""
    rp, ci, llr = got
    rrp, rci, rllr = ref.to_csr()
    assert rp.shape == rrp.shape
    assert np.array_equal(rp, rrp), f"row lengths differ at rows {np.nonzero(np.diff(rp) != np.diff(rrp))[0][:10]}"
    if llr.size:
        assert np.abs(llr - rllr).max() <= tol, f"max |dLLR| = {np.abs(llr - rllr).max()}"
    if np.array_equal(ci, rci):
        return rp.size - 1, 0
    assert not exact_ids, "ids differ"
    bad_rows = np.unique(np.searchsorted(rp, np.nonzero(ci != rci)[0], side="right") - 1)
    k = ref.k
    for r in bad_rows:
        s, e = rp[r], rp[r + 1]
        g = dict(zip(ci[s:e].tolist(), llr[s:e].tolist()))
        o = dict(zip(rci[s:e].tolist(), rllr[s:e].tolist()))
        kth = rllr[e - 1]
        for j, v in g.items():
            if j in o:
                assert abs(o[j] - v) <= tol, f"row {r} col {j}: {v} vs {o[j]}"
            else:
                assert e - s == k and abs(v - kth) <= tol, f"row {r}: col {j} (llr {v}) not in the oracle row and not a k-boundary tie"
        for j, v in o.items():
            if j not in g:
                assert e - s == k and abs(v - llr[e - 1]) <= tol, f"row {r}: oracle col {j} (llr {v}) missing and not a k-boundary tie"
        # order: positions may only swap inside groups of (near-)equal score
        for a, b in zip(ci[s:e], rci[s:e]):
            if a != b:
                assert abs(g[a] - o[b]) <= tol
    return rp.size - 1, len(bad_rows)


def run_device(sess, mats, params, seed, mode=0, item_lo=0, item_hi=None):
    dev = sess.device
    out = D.cross_occurrence_device(sess, [to_dev(m, dev) for m in mats], to_params(params), seed, mode, item_lo, item_hi)
    sess.synchronize()
    return out


def compare_with_oracle(sess, mats, params, seed, mode=0, item_lo=0, item_hi=None, exact_ids=False):
    out = run_device(sess, mats, params, seed, mode, item_lo, item_hi)
    ref = O.cross_occurrence_downsampled(mats, params, seed, mode, 1, item_lo, item_hi)
    stats = []
    for o, r in zip(out, ref):
        st = o.stats.cpu().numpy()
        assert int(st[0]) == r.pairs, f"pairs {int(st[0])} vs oracle {r.pairs}"
        assert int(st[1 + 4 * 7]) == 0, "LDS accumulator overflow reported"
        n, ties = check_indicators(o.to_host(), r, exact_ids=exact_ids)
        stats.append((st.copy(), n, ties))
    return out, ref, stats


class RanksOfAJob:
    """The per-rank outputs of ONE event type of a multi-rank build (disjoint, consecutive item ranges) seen as one indicator matrix."""

    def __init__(self, parts):
        assert parts[0].item_lo == 0 and all(a.item_hi == b.item_lo for a, b in zip(parts, parts[1:])), "the ranks' item ranges do not tile the items"
        self.parts = parts
        self.stats = torch.stack([p.stats.cpu() for p in parts]).sum(0)

    def to_host(self):
        hosts = [p.to_host() for p in self.parts]
        lens = np.concatenate([np.diff(h[0]) for h in hosts])
        rp = np.zeros(lens.size + 1, np.int64)
        np.cumsum(lens, out=rp[1:])
        return rp, np.concatenate([h[1] for h in hosts]), np.concatenate([h[2] for h in hosts])

    def check_sampled(self, b):
        for p in self.parts:
            check_sampled_rows(p, b)


def check_sampled_rows(o, b):
    """The down-sampled B a GPU multiplied with, after an exchange: every row either whole (bit for bit) or -- row-filtered exchange: no
    item of this GPU's range among the user's primary items -- empty; the library's own total is the whole matrix's."""
    rp_g = o.sampled_row_ptr.cpu().numpy()
    lens_g, lens_f = np.diff(rp_g), np.diff(b.row_ptr)
    assert lens_g.shape == lens_f.shape and np.all((lens_g == lens_f) | (lens_g == 0)), "a down-sampled row arrived neither whole nor empty"
    held = lens_g > 0
    ci_g = o.sampled_col_idx[: int(rp_g[-1])].cpu().numpy()
    assert np.array_equal(ci_g, b.col_idx[np.repeat(held, lens_f)]), "down-sampled col_idx of the rows this GPU holds differs"
    assert o.sampled_nnz_total in (-1, b.nnz), (o.sampled_nnz_total, b.nnz)


def shard_rows(dev_mats, W):
    """User-range shards of device-resident matrices: [event type][rank] (views; row_ptr re-based to 0)."""
    n = dev_mats[0].n_rows
    cuts = [n * g // W for g in range(W + 1)]
    out = []
    for m in dev_mats:
        e = [int(m.row_ptr[c].item()) for c in cuts]
        out.append([D.DevCsr(cuts[g + 1] - cuts[g], m.n_cols, m.row_ptr[cuts[g]:cuts[g + 1] + 1] - e[g], m.col_idx[e[g]:max(e[g + 1], e[g] + 1)], e[g + 1] - e[g])
                    for g in range(W)])
    return out, cuts


def compare_with_oracle_large(sess, mats, params, seed, mode=0, threads=None, dev_mats=None, via_context=False, flags=0, also=()):
    """compare_with_oracle for workloads of 10^8..10^9 pairs: the C oracle runs on every host core, one event type at a
    time (its strided outputs are freed before the next), and the down-sampled matrices themselves -- row_ptr AND
    col_idx -- are compared bit for bit before the indicator rows (every row) are.  `dev_mats`: the same matrices already
    resident in HBM (inputs generated on the device).  `via_context`: run the build through urcco_context_build_device (the
    entry point bench.py times) instead of the session-level stage calls.  `also`: (label, per-event outputs) of OTHER builds of the same
    job -- the exchange route, the ranks of an emulated multi-rank build -- checked against the same oracle pass (the oracle is what a
    test at this size costs).  Returns per event (stats vector, rows needing the k-boundary tie rule)."""
    threads = threads or min(os.cpu_count() or 1, O.lib().orc_max_threads())
    ctx = None
    if dev_mats is None:
        dev_mats = [to_dev(m, sess.device) for m in mats]
    if via_context:
        ctx = D.Context(sess.device, sess.lib, 1, flags, mode)
        out = D.cross_occurrence_context(ctx, dev_mats, to_params(params), seed)
    else:
        out = D.cross_occurrence_device(sess, dev_mats, to_params(params), seed, mode)
        sess.synchronize()
    try:
        a = O.downsample(mats[0], O.column_counts(mats[0]), seed, params[0].max_elements_per_row, mode)
        cnt_a = O.column_counts(a)
        a_cp, a_ri = O.transpose(a)
        res = []
        for d, (m, p, o) in enumerate(zip(mats, params, out)):
            b = a if d == 0 else O.downsample(m, O.column_counts(m), seed, p.max_elements_per_row, mode)
            cnt_b = cnt_a if d == 0 else O.column_counts(b)
            assert np.array_equal(o.sampled_row_ptr.cpu().numpy(), b.row_ptr), f"event {d}: down-sampled row_ptr differs"
            assert np.array_equal(o.sampled_col_idx[:b.nnz].cpu().numpy(), b.col_idx), f"event {d}: down-sampled col_idx differs"
            ref = O.cco_rows(a_cp, a_ri, b, cnt_a, cnt_b, mats[0].n_rows, d == 0, p.max_interesting_elements, p.min_llr, 0, None, threads)
            st = o.stats.cpu().numpy()
            assert int(st[0]) == ref.pairs, f"event {d}: pairs {int(st[0])} vs oracle {ref.pairs}"
            assert int(st[1 + 4 * 7]) == 0, "LDS accumulator overflow reported"
            _, ties = check_indicators(o.to_host(), ref)
            res.append((st.copy(), ties))
            for label, outs in also:
                x = outs[d]
                xs = x.stats.cpu().numpy()
                assert int(xs[0]) == ref.pairs, f"{label}, event {d}: pairs {int(xs[0])} vs oracle {ref.pairs}"
                assert int(xs[1 + 4 * 7]) == 0, f"{label}: LDS accumulator overflow reported"
                if hasattr(x, "check_sampled"):
                    x.check_sampled(b)
                else:
                    check_sampled_rows(x, b)
                check_indicators(x.to_host(), ref)
            del ref, b
    finally:
        if ctx is not None:
            out = None
            ctx.close()
    return out, res


def device_generated(cfg, device):
    """Inputs of a BASELINE configuration generated ON THE GPU (synth.generate_device) and mirrored to the host for the
    oracle: (device matrices, host matrices)."""
    from universal_recommender_amd import synth
    dev_mats, mats = [], []
    for (_, nc, rp, ci) in synth.generate_device(cfg, device):
        nnz = int(rp[-1].item())
        dev_mats.append(D.DevCsr(cfg.n_users, nc, rp, ci, nnz))
        mats.append(O.Csr(cfg.n_users, nc, rp.cpu().numpy(), ci.cpu().numpy()))
    return dev_mats, mats


def sort_rows(got):
    """(row_ptr, col_idx, llr) with every row re-ordered to the canonical (llr desc, col asc) -- for outputs produced under
    URCCO_FLAG_UNORDERED_ROWS, whose rows carry the right SET in arbitrary order."""
    rp, ci, llr = got
    rows = np.repeat(np.arange(rp.size - 1), np.diff(rp))
    order = np.lexsort((ci, -llr, rows))
    return rp, ci[order], llr[order]
