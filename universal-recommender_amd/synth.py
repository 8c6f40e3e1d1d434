"""Synthetic user-event streams for BASELINE.json configs 3-5, already in the integer-indexed CSR form
Preparator.prepare produces (ids are dense, every user has >= 1 primary event).

Generator: numpy Generator(PCG64(seed)); user degree d_u = 1 + Poisson(lambda_event); items i.i.d. from a bounded
Zipf(s) over ranks 1..n_items (inverse-CDF), de-duplicated inside a user; rank -> id through a fixed random
permutation so hot items are not contiguous.  Config 5 mixes a uniform draw from the top 0.1 % (prob 0.4) with the
Zipf tail and multiplies 1 % of the users' degrees by 50.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np


@dataclass
class EventSpec:
    name: str
    lam: float                 # d_u = 1 + Poisson(lam)   (secondary events: Poisson(lam), may be 0)
    n_items: int
    zipf_s: float = 1.0


@dataclass
class SynthConfig:
    name: str
    n_users: int
    events: List[EventSpec]
    seed: int
    skew_top_frac: float = 0.0   # config 5: fraction of items forming the hot head
    skew_top_prob: float = 0.0   # probability mass drawn uniformly from the head
    heavy_user_frac: float = 0.0
    heavy_user_mult: int = 1


def config3(scale: float = 1.0) -> SynthConfig:
    nu, ni = int(1_000_000 * scale), max(int(200_000 * scale), 16)
    return SynthConfig("config3-1Mx200K-zipf1.0-3events", nu, [
        EventSpec("purchase", 9, ni), EventSpec("view", 39, ni), EventSpec("category-pref", 2, max(int(2000 * min(scale * 10, 1.0)), 8))],
        seed=20260925 + 3)


def config4(scale: float = 1.0, item_scale: Optional[float] = None) -> SynthConfig:
    """scale shrinks the users; the item spaces follow unless item_scale says otherwise (item_scale = 1.0 keeps the
    2M-wide column spaces of the full configuration -- what the packed accumulator keys and bucket counts depend on)."""
    isc = scale if item_scale is None else item_scale
    nu, ni = int(10_000_000 * scale), max(int(2_000_000 * isc), 16)
    return SynthConfig("config4-10Mx2M-zipf1.0-5events", nu, [
        EventSpec("purchase", 9, ni), EventSpec("view", 39, ni), EventSpec("add-to-cart", 14, ni),
        EventSpec("search", 19, max(int(200_000 * isc), 16)), EventSpec("category-pref", 2, max(int(2000 * min(isc * 10, 1.0)), 8))],
        seed=20260925 + 4)


def config5(scale: float = 1.0, item_scale: Optional[float] = None) -> SynthConfig:
    c = config4(scale, item_scale)
    c.name = "config5-10Mx2M-skewed-5events"
    c.seed = 20260925 + 5
    c.skew_top_frac, c.skew_top_prob = 0.001, 0.40
    c.heavy_user_frac, c.heavy_user_mult = 0.01, 50
    return c


def _zipf_cdf(n: int, s: float) -> np.ndarray:
    w = 1.0 / np.power(np.arange(1, n + 1, dtype=np.float64), s)
    c = np.cumsum(w)
    return c / c[-1]


def generate_event(rng: np.random.Generator, cfg: SynthConfig, ev: EventSpec, primary: bool, user_lo: int = 0,
                   user_hi: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
    """CSR (row_ptr int64, col_idx int32) of one event type for users [user_lo, user_hi)."""
    user_hi = cfg.n_users if user_hi is None else user_hi
    n = user_hi - user_lo
    deg = rng.poisson(ev.lam, n).astype(np.int64) + (1 if primary else 0)
    if cfg.heavy_user_frac > 0:
        heavy = rng.random(n) < cfg.heavy_user_frac
        deg[heavy] *= cfg.heavy_user_mult
    total = int(deg.sum())
    u = rng.random(total)
    if cfg.skew_top_prob > 0 and ev.n_items >= 1000:
        n_top = max(int(ev.n_items * cfg.skew_top_frac), 1)
        from_top = rng.random(total) < cfg.skew_top_prob
        cdf = _zipf_cdf(ev.n_items - n_top, ev.zipf_s)
        ranks = np.where(from_top, (u * n_top).astype(np.int64), n_top + np.searchsorted(cdf, u, side="left"))
    else:
        cdf = _zipf_cdf(ev.n_items, ev.zipf_s)
        ranks = np.searchsorted(cdf, u, side="left")
    ranks = np.minimum(ranks, ev.n_items - 1)
    perm = np.random.Generator(np.random.PCG64(cfg.seed * 1000003 + ev.n_items)).permutation(ev.n_items)
    items = perm[ranks].astype(np.int64)
    users = np.repeat(np.arange(n, dtype=np.int64), deg)
    key = np.unique(users * ev.n_items + items)          # sort by (user, item) and de-duplicate
    users_u = key // ev.n_items
    col = (key - users_u * ev.n_items).astype(np.int32)
    row_ptr = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(users_u, minlength=n), out=row_ptr[1:])
    return row_ptr, col


def generate(cfg: SynthConfig, user_lo: int = 0, user_hi: Optional[int] = None):
    """List of (event name, n_cols, row_ptr, col_idx) for users [user_lo, user_hi); the stream of a user shard depends
    only on (cfg.seed, event index, user_lo), so ranks can generate their own shards."""
    out = []
    for e, ev in enumerate(cfg.events):
        rng = np.random.Generator(np.random.PCG64([cfg.seed, e, user_lo]))
        rp, ci = generate_event(rng, cfg, ev, primary=(e == 0), user_lo=user_lo, user_hi=user_hi)
        out.append((ev.name, ev.n_items, rp, ci))
    return out


def generate_device(cfg: SynthConfig, device, user_lo: int = 0, user_hi: Optional[int] = None):
    """The same generator evaluated ON THE GPU with torch (device RNG: Philox, so the streams differ from `generate`'s
    PCG64 -- same distributions, different draws): list of (event name, n_cols, row_ptr int64 tensor, col_idx int32
    tensor) resident in HBM.  For the 10M x 2M configurations, where the numpy path needs minutes of host time per build
    of the inputs."""
    import torch
    user_hi = cfg.n_users if user_hi is None else user_hi
    n = user_hi - user_lo
    out = []
    for e, ev in enumerate(cfg.events):
        g = torch.Generator(device=device)
        g.manual_seed(int(cfg.seed) * 1000003 + e * 7919 + user_lo)
        deg = torch.poisson(torch.full((n,), float(ev.lam), device=device, dtype=torch.float32), generator=g).to(torch.int64) + (1 if e == 0 else 0)
        if cfg.heavy_user_frac > 0:
            heavy = torch.rand(n, device=device, generator=g) < cfg.heavy_user_frac
            deg = torch.where(heavy, deg * cfg.heavy_user_mult, deg)
        total = int(deg.sum().item())
        u = torch.rand(total, device=device, dtype=torch.float64, generator=g)
        if cfg.skew_top_prob > 0 and ev.n_items >= 1000:
            n_top = max(int(ev.n_items * cfg.skew_top_frac), 1)
            from_top = torch.rand(total, device=device, generator=g) < cfg.skew_top_prob
            cdf = torch.from_numpy(_zipf_cdf(ev.n_items - n_top, ev.zipf_s)).to(device)
            ranks = torch.where(from_top, (u * n_top).to(torch.int64), n_top + torch.searchsorted(cdf, u))
        else:
            cdf = torch.from_numpy(_zipf_cdf(ev.n_items, ev.zipf_s)).to(device)
            ranks = torch.searchsorted(cdf, u)
        del u
        ranks.clamp_(max=ev.n_items - 1)
        perm = torch.from_numpy(np.random.Generator(np.random.PCG64(cfg.seed * 1000003 + ev.n_items)).permutation(ev.n_items)).to(device)
        key = torch.repeat_interleave(torch.arange(n, device=device, dtype=torch.int64), deg) * ev.n_items + perm[ranks]
        del ranks, perm
        key = torch.unique(key)                               # sorted by (user, item), duplicates collapsed
        users_u = torch.div(key, ev.n_items, rounding_mode="floor")
        col = (key - users_u * ev.n_items).to(torch.int32)
        row_ptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
        torch.cumsum(torch.bincount(users_u, minlength=n), 0, out=row_ptr[1:])
        del key, users_u
        out.append((ev.name, ev.n_items, row_ptr, col))
    return out
