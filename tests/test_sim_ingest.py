"""Device-side Preparator (universal_recommender_amd/ingest.py: dictionaries + CSR builder kernels) on the host simulator
against the oracle's restatement of Preparator.prepare: same dictionaries (first-appearance order, minEventsPerUser on
RAW primary events), same binary matrices, bit for bit."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import cco_oracle as PO


def key64(s: str) -> int:
    v = int.from_bytes(hashlib.blake2b(s.encode(), digest_size=8).digest(), "little", signed=False)
    if v == (1 << 64) - 1:      # reserved "empty slot" value
        v = 0
    return v - (1 << 64) if v >= (1 << 63) else v   # as the int64 bit pattern


def keys_tensor(strings, device):
    return torch.tensor([key64(s) for s in strings], dtype=torch.int64, device=device)


def check_prepare(sess, actions, min_events):
    from universal_recommender_amd import ingest
    dev = sess.device
    ref = PO.prepare(actions, min_events)
    nonempty = [(n, e) for (n, e) in actions if len(e) > 0]           # DataSource.scala:89 (the host drops them)
    dev_actions = [(n, keys_tensor([u for u, _ in e], dev), keys_tensor([i for _, i in e], dev)) for (n, e) in nonempty]
    got = ingest.prepare_device(sess, dev_actions, min_events)
    sess.synchronize()
    assert len(got.events) == len(ref)
    users = ref[0][1].row_ids
    ufp = got.user_first_pos.cpu().numpy()
    assert ufp.shape[0] == len(users)
    primary_users = [u for u, _ in nonempty[0][1]]
    assert [primary_users[p] for p in ufp] == users.keys            # id -> string through the first-occurrence positions
    for (name, e), ev, (rname, ids) in zip(nonempty, got.events, ref):
        assert ev.name == rname == name
        m = ev.matrix
        assert m.n_rows == ids.nrow and m.n_cols == ids.ncol
        items = [i for _, i in e]
        ifp = ev.item_first_pos.cpu().numpy()
        assert [items[p] for p in ifp] == ids.column_ids.keys
        rp = m.row_ptr.cpu().numpy()
        ci = m.col_idx.cpu().numpy()
        lens = [len(r) for r in ids.rows]
        assert np.array_equal(np.diff(rp), np.asarray(lens, dtype=np.int64))
        assert m.nnz_bound == sum(lens)
        flat = [c for r in ids.rows for c in r]
        assert np.array_equal(ci[: len(flat)], np.asarray(flat, dtype=np.int32))
    return got


def random_actions(rng, n_users, n_items, n_events, n_types=3, user_pool_extra=0):
    acts = []
    for d in range(n_types):
        nu = n_users + (user_pool_extra if d > 0 else 0)           # secondaries see users the primary never had
        us = rng.zipf(1.3, size=n_events[d]) % nu
        it = rng.zipf(1.2, size=n_events[d]) % n_items[d]
        acts.append((f"ev{d}", [(f"u{u}", f"i{d}_{i}") for u, i in zip(us, it)]))
    return acts


@pytest.mark.parametrize("min_events", [None, 1, 2, 5])
def test_prepare_matches_the_oracle(sim_session, min_events):
    rng = np.random.default_rng(3 if min_events is None else 10 + min_events)
    acts = random_actions(rng, 700, [300, 900, 12], [4000, 9000, 1500], user_pool_extra=150)
    check_prepare(sim_session, acts, min_events)


def test_prepare_handmade_and_degenerate_streams(sim_session):
    acts = [("purchase", [("u1", "a"), ("u2", "a"), ("u1", "b"), ("u1", "a"), ("u3", "c"), ("u2", "b")]),
            ("view", [("u9", "x"), ("u1", "x"), ("u1", "x"), ("u3", "y"), ("u2", "x")]),
            ("empty", []),
            ("pref", [("u4", "p")])]                                  # no surviving event: 0 columns, all rows empty
    check_prepare(sim_session, acts, None)
    check_prepare(sim_session, acts, 2)                               # u3 has one purchase: dropped, item c vanishes
    check_prepare(sim_session, [("only", [("u", "i")] * 5)], 3)
    got = check_prepare(sim_session, [("only", [("u", "i")] * 2)], 3)  # nobody qualifies: empty dictionaries
    assert got.user_first_pos.numel() == 0


def test_prepare_long_rows_take_the_block_sorts(sim_session):
    """One user with > 4096 raw events (global-memory bitonic), one with a few hundred (LDS bitonic), many duplicates."""
    rng = np.random.default_rng(5)
    ev = [("heavy", f"i{int(x)}") for x in rng.integers(0, 6000, 9000)]
    ev += [("medium", f"i{int(x)}") for x in rng.integers(0, 300, 700)]
    ev += [(f"u{int(u)}", f"i{int(x)}") for u, x in zip(rng.integers(0, 200, 3000), rng.integers(0, 6000, 3000))]
    order = rng.permutation(len(ev))
    acts = [("purchase", [ev[k] for k in order])]
    check_prepare(sim_session, acts, None)
