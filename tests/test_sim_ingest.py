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


def _same_prepared(a, b):
    assert [n for n, _ in a.actions] == [n for n, _ in b.actions]
    for (_, x), (_, y) in zip(a.actions, b.actions):
        assert x.rowIDs.keys == y.rowIDs.keys and x.columnIDs.keys == y.columnIDs.keys
        assert np.array_equal(x.row_ptr, y.row_ptr) and np.array_equal(x.col_idx, y.col_idx)


def test_preparator_mirror_device_equals_host(sim_session):
    """universal_recommender_amd.Preparator: prepare_on_device (ingest kernels) == prepare (numpy), on the reference's
    handmade events with its minEventsPerUser = 3 and on a random stream."""
    import json
    import os
    from universal_recommender_amd.data_source import DataSource, DataSourceParams, TrainingData
    from universal_recommender_amd.preparator import Preparator
    doc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "handmade.json")))
    lines = [",".join(e) for e in doc["events"]]
    dsp = doc["datasource_params"]
    td = DataSource(DataSourceParams(dsp["appName"], dsp["eventNames"], None, dsp["minEventsPerUser"])).readTraining(lines)
    host = Preparator().prepare(td)
    _same_prepared(Preparator().prepare_on_device(td, sim_session), host)
    assert [(d.row_ptr.size - 1, d.columnIDs.size, d.col_idx.size) for _, d in host.actions] == [(3, 6, 11), (3, 4, 10), (3, 2, 5)]
    rng = np.random.default_rng(41)
    acts = random_actions(rng, 300, [90, 400, 7], [2500, 6000, 800], user_pool_extra=40)
    for min_events in (None, 1, 3):
        td = TrainingData(acts, {}, min_events)
        _same_prepared(Preparator().prepare_on_device(td, sim_session), Preparator().prepare(td))


def test_events_to_model_without_leaving_the_device(sim_session, sim_lib):
    """URAlgorithm.train_events_on_device (ingest kernels -> device matrices -> CCO build) builds the same model documents
    as the host-boundary path and as the oracle, on the reference's handmade and item-set goldens."""
    import json
    import os
    from universal_recommender_amd.data_source import DataSource, DataSourceParams
    from universal_recommender_amd.preparator import Preparator
    from universal_recommender_amd.ur_algorithm import URAlgorithm, URAlgorithmParams, toStringMap
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for name in ("handmade.json", "item_sets.json"):
        doc = json.load(open(os.path.join(golden, name)))
        lines = [",".join(e) for e in doc["events"]]
        engine = {"datasource": {"params": doc["datasource_params"]}, "algorithms": [{"name": "ur", "params": doc["algorithm_params"]}]}
        td = DataSource(DataSourceParams.from_engine_json(engine)).readTraining(lines)
        ap = URAlgorithmParams.from_engine_json(engine)
        ap.seed = 1
        algo = URAlgorithm(ap, library=sim_lib)

        def docs(result):
            model = {}
            for ev, ind in result:
                for item, m in toStringMap(ind, ev).items():
                    model.setdefault(item, {}).update(m)
            return model
        on_device = algo.train_events_on_device(td, sim_session)
        via_host = algo.train(Preparator().prepare(td)).coocurrenceMatrices
        assert docs(on_device) == docs(via_host)
        for (_, a), (_, b) in zip(on_device, via_host):
            assert np.array_equal(a.row_ptr, b.row_ptr) and np.array_equal(a.col_idx, b.col_idx) and np.array_equal(a.values, b.values)
        by_event = {}
        for u, e, i in doc["events"]:
            by_event.setdefault(e, []).append((u, i))
        names = doc["datasource_params"]["eventNames"]
        prepared = PO.prepare(PO.split_actions(by_event, names), doc["datasource_params"].get("minEventsPerUser"))
        ref = {}
        for ev, ind in PO.calc_all(prepared, {**doc["algorithm_params"], "seed": 1}):
            for item, m in PO.to_string_map(ev, ind).items():
                ref.setdefault(item, {}).update(m)
        assert docs(on_device) == ref


def test_reference_downsample_fixture_user_filter(sim_session, sim_lib):
    """data/sample-downsamplable-data.txt with examples/handmade-engine-downsample.json (minEventsPerUser = 6; the reference
    ships no expected output for it): u3 and u4 have 5 purchases and are dropped, the items only they bought (p11, p12)
    leave the purchase dictionary, their views are dropped too.  Oracle == host mirror == device ingest, and the model
    built from the device matrices equals the oracle's."""
    import json
    import os
    from universal_recommender_amd.data_source import DataSource, DataSourceParams
    from universal_recommender_amd.preparator import Preparator
    from universal_recommender_amd.ur_algorithm import URAlgorithm, URAlgorithmParams, toStringMap
    doc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "downsample.json")))
    engine = {"datasource": {"params": doc["datasource_params"]}, "algorithms": [{"name": "ur", "params": doc["algorithm_params"]}]}
    td = DataSource(DataSourceParams.from_engine_json(engine)).readTraining([",".join(e) for e in doc["events"]])
    assert td.minEventsPerUser == 6
    by_event = {}
    for u, e, i in doc["events"]:
        by_event.setdefault(e, []).append((u, i))
    prepared = PO.prepare(PO.split_actions(by_event, doc["datasource_params"]["eventNames"]), 6)
    users = prepared[0][1].row_ids.keys
    assert "u3" not in users and "u4" not in users and {"u1", "u2", "u5"} <= set(users)
    assert "p11" not in prepared[0][1].column_ids.keys and "p12" not in prepared[0][1].column_ids.keys
    host = Preparator().prepare(td)
    _same_prepared(Preparator().prepare_on_device(td, sim_session), host)
    for (_, ids), (_, ref) in zip(host.actions, prepared):
        assert ids.rowIDs.keys == ref.row_ids.keys and ids.columnIDs.keys == ref.column_ids.keys
        assert [list(ids.col_idx[ids.row_ptr[r]:ids.row_ptr[r + 1]]) for r in range(ref.nrow)] == ref.rows
    ap = URAlgorithmParams.from_engine_json(engine)
    ap.seed = 7
    model = {}
    for ev, ind in URAlgorithm(ap, library=sim_lib).train_events_on_device(td, sim_session):
        for item, m in toStringMap(ind, ev).items():
            model.setdefault(item, {}).update(m)
    ref_model = {}
    for ev, ind in PO.calc_all(prepared, {**doc["algorithm_params"], "seed": 7}):
        for item, m in PO.to_string_map(ev, ind).items():
            ref_model.setdefault(item, {}).update(m)
    assert model == ref_model


def test_native_string_hash_known_answers_and_collision_check(sim_session, sim_lib):
    """urcco_hash_strings = XXH64 (known answers of the public algorithm, seed 0 and the check seed); a dictionary in which
    two different strings share the 64-bit key is detected through the second hash (urcco_dev_dictionary_verify)."""
    import torch
    from universal_recommender_amd import ingest
    from universal_recommender_amd.preparator import CHECK_SEED, hash_keys
    strs = ["", "a", "user-1", "Iphone 6", "x" * 40]
    got = hash_keys(strs, 0, sim_lib).view(np.uint64).tolist()
    assert got == [0xef46db3751d8e999, 0xd24ec4f1a98c6e5b, 0xa173746b114c6be8, 0x4ae1af7e7ed5ca3b, 0x926f564e1b3e18d5]
    got2 = hash_keys(strs, CHECK_SEED, sim_lib).view(np.uint64).tolist()
    assert got2 == [0xc4349fc93c010000, 0x9a7c6d2ea45568c9, 0xf45ce0957975e277, 0x2d02c0058280b236, 0x5c39218a651d07bb]
    try:
        import xxhash
        assert got == [xxhash.xxh64_intdigest(s.encode()) for s in strs]
        assert got2 == [xxhash.xxh64_intdigest(s.encode(), seed=CHECK_SEED) for s in strs]
    except ImportError:
        pass
    dev = sim_session.device
    t = lambda x: torch.from_numpy(np.asarray(x, np.int64)).to(dev)
    users, items = [f"u{i % 7}" for i in range(40)], [f"i{i % 11}" for i in range(40)]
    uk, ik = hash_keys(users, 0, sim_lib), hash_keys(items, 0, sim_lib)
    uc, ic = hash_keys(users, CHECK_SEED, sim_lib), hash_keys(items, CHECK_SEED, sim_lib)
    ok = ingest.prepare_device(sim_session, [("buy", t(uk), t(ik), t(uc), t(ic))], 1)
    assert ok.user_first_pos.numel() == 7 and ok.events[0].matrix.n_cols == 11
    ik_bad = ik.copy()
    ik_bad[np.asarray(items) == "i3"] = ik[items.index("i5")]          # "i3" and "i5" now share a key, their check keys differ
    with pytest.raises(ingest.HashCollision):
        ingest.prepare_device(sim_session, [("buy", t(uk), t(ik_bad), t(uc), t(ic))], 1)
    uk_bad = uk.copy()
    uk_bad[np.asarray(users) == "u1"] = uk[users.index("u2")]
    with pytest.raises(ingest.HashCollision):
        ingest.prepare_device(sim_session, [("buy", t(uk_bad), t(ik), t(uc), t(ic))], 1)
    # a SECONDARY event type whose user "v9" shares its 64-bit key with the primary's "u2" (different strings, different check
    # keys): merged into u2 silently before; detected through the check keys of the primary's first occurrences now
    users2 = [f"u{i % 5}" for i in range(20)] + ["v9"] * 3
    items2 = [f"j{i % 4}" for i in range(23)]
    uk2, ik2 = hash_keys(users2, 0, sim_lib), hash_keys(items2, 0, sim_lib)
    uc2, ic2 = hash_keys(users2, CHECK_SEED, sim_lib), hash_keys(items2, CHECK_SEED, sim_lib)
    good = ingest.prepare_device(sim_session, [("buy", t(uk), t(ik), t(uc), t(ic)), ("view", t(uk2), t(ik2), t(uc2), t(ic2))], 1)
    assert good.events[1].matrix.n_cols == 4
    uk2_bad = uk2.copy()
    uk2_bad[np.asarray(users2) == "v9"] = uk[users.index("u2")]
    with pytest.raises(ingest.HashCollision):
        ingest.prepare_device(sim_session, [("buy", t(uk), t(ik), t(uc), t(ic)), ("view", t(uk2_bad), t(ik2), t(uc2), t(ic2))], 1)
