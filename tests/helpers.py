"""Shared test helpers: seeded matrix generators, host<->device conversion, the tie-aware parity checker."""
from __future__ import annotations

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


def to_dev(m: O.Csr, device) -> D.DevCsr:
    ci = m.col_idx if m.nnz else np.zeros(1, np.int32)
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
