"""PARITY AT SCALE (-m gpu): the workloads the BASELINE configs name, compared with the C oracle ROW BY ROW -- every
indicator row of every event type, and the down-sampled matrices themselves (row_ptr and col_idx) -- not through
samples.  The 10M x 2M configurations keep their 2M-wide item spaces (21-bit packed keys -> 11-bit packed counts, 245
column buckets); the users are scaled so that a test stays within minutes of host time for the oracle."""
import numpy as np
import pytest
import torch

from helpers import compare_with_oracle_large
from oracle import c_oracle as O

pytestmark = pytest.mark.gpu


def P(max_rows=500, k=50, min_llr=None):
    return O.DatasetParams(max_rows, k, min_llr)


def _mats(cfg):
    from universal_recommender_amd import synth
    return [O.Csr(cfg.n_users, nc, rp, ci) for (_, nc, rp, ci) in synth.generate(cfg)]


def test_full_config3_every_row(gpu_session):
    """BASELINE config 3 at FULL size (1M x 200K, 3 events; the bench workload): all 600K indicator rows, the three
    down-sampled matrices bit for bit, pairs."""
    from universal_recommender_amd import synth
    mats = _mats(synth.config3(1.0))
    _, res = compare_with_oracle_large(gpu_session, mats, [P(), P(), P()], 20260925)
    assert sum(int(st[0]) for st, _ in res) == 90668283        # the pairs figure bench.py reports for this seed


def test_config4_quarter_scale_full_item_space(gpu_session):
    """BASELINE config 4 with 2.5M users (1/4) x the full 2M / 2M / 2M / 200K / 2K item spaces, 5 event types."""
    from universal_recommender_amd import synth
    cfg = synth.config4(0.25, item_scale=1.0)
    mats = _mats(cfg)
    assert mats[0].n_cols == 2_000_000 and len(mats) == 5
    _, res = compare_with_oracle_large(gpu_session, mats, [P()] * 5, 4)
    rows_by_bin = np.sum([st[1:8] for st, _ in res], axis=0)
    assert rows_by_bin[0] > 0 and rows_by_bin[1] > 0 and rows_by_bin[5] + rows_by_bin[6] > 0, rows_by_bin


def test_config5_tenth_scale_full_item_space_skew(gpu_session):
    """BASELINE config 5 (hot head: top 0.1 % of the items draw 40 % of the interactions; 1 % heavy users) with 1M users
    (1/10) x the full item spaces, `indicators` form.  maxItemsPerUser = 500 drops the heavy users' rows (Int/Int row
    rate); one event type runs with maxItemsPerUser = 3000, which leaves columns with more than 2047 interactions --
    beyond the 11-bit packed count of a 2M-wide column space -- so those rows must take the global accumulator."""
    from universal_recommender_amd import synth
    cfg = synth.config5(0.1, item_scale=1.0)
    mats = _mats(cfg)
    assert max(int(np.diff(m.row_ptr).max()) for m in mats) > 500
    params = [P(3000, 50), P(500, 50), P(500, 20, 2.0), P(500, 50), P(500, 50)]
    out, res = compare_with_oracle_large(gpu_session, mats, params, 5)
    rows_by_bin = np.sum([st[1:8] for st, _ in res], axis=0)
    assert rows_by_bin[6] > 0, rows_by_bin                      # global-accumulator class used
    raw_len = np.diff(mats[1].row_ptr)
    kept_len = np.diff(out[1].sampled_row_ptr.cpu().numpy())
    assert np.all(kept_len[raw_len > 500] == 0) and (raw_len > 500).sum() > 1000    # maxElementsPerRow drops
    # fractional row rate on the two heaviest event types
    compare_with_oracle_large(gpu_session, mats[:2], [P(500, 50), P(500, 50)], 5, mode=1)


def test_full_config4_every_row(gpu_session):
    """BASELINE config 4 at FULL size (10M users x 2M / 2M / 2M / 200K / 2K items, 5 event types, ~784M raw interactions,
    ~1.06G cooccurrence pairs) -- the workload bench.py's default line is quoted on -- through the entry point it times
    (urcco_context_build_device): the five down-sampled matrices bit for bit, every one of the 10M indicator rows.  Inputs
    are generated on the GPU and mirrored to the host for the oracle (all host cores)."""
    from helpers import device_generated
    from universal_recommender_amd import synth
    cfg = synth.config4(1.0)
    dev_mats, mats = device_generated(cfg, gpu_session.device)
    assert mats[0].n_rows == 10_000_000 and mats[0].n_cols == 2_000_000 and len(mats) == 5
    # the same job through the EXCHANGE route at full size -- URCCO_FLAG_FORCE_EXCHANGE: a one-rank RCCL communicator, count
    # all-reduces, work-balanced ranges, own-shard transposition + fragment merge, need masks, packing, the row-filtered all-to-all-v, the
    # fused expand of the sharded path -- against the same oracle pass, every row
    from helpers import to_params
    from universal_recommender_amd import _lib
    from universal_recommender_amd import device as D
    ctx_x = D.Context(gpu_session.device, gpu_session.lib, 1, _lib.FLAG_FORCE_EXCHANGE, 0)
    try:
        out_x = D.cross_occurrence_context(ctx_x, dev_mats, to_params([P()] * 5), 20260925)
        _, res = compare_with_oracle_large(gpu_session, mats, [P()] * 5, 20260925, dev_mats=dev_mats, via_context=True, also=[("exchange route", out_x)])
    finally:
        out_x = None
        ctx_x.close()
    rows_by_bin = np.sum([st[1:8] for st, _ in res], axis=0)
    assert sum(int(st[0]) for st, _ in res) > 1_000_000_000 and rows_by_bin[:6].min() > 0, rows_by_bin


def test_full_config5_every_row(gpu_session):
    """BASELINE config 5 at FULL size (config 4's shape with the hot head -- top 0.1 % of the items draw 40 % of the
    interactions -- and 1 % heavy users x50), engine.json defaults as bench.py --workload config5 runs it: the heavy
    users' rows are dropped by maxItemsPerUser (Int / Int row rate), the hot items' rows are too heavy for any single-pass
    LDS accumulator (the multi-pass class)."""
    from helpers import device_generated
    from universal_recommender_amd import synth
    cfg = synth.config5(1.0)
    dev_mats, mats = device_generated(cfg, gpu_session.device)
    assert max(int(np.diff(m.row_ptr).max()) for m in mats) > 500
    # the same job as EIGHT ranks (BASELINE config 5: "8 x MI355X") on this one GPU -- URCCO_FLAG_EMULATE_RANKS, the real
    # 8-rank build with its collectives looped back on the device -- every rank's rows of the skewed catalogue against the same oracle pass
    from helpers import RanksOfAJob, shard_rows, to_params
    from universal_recommender_amd import _lib, sharded
    from universal_recommender_amd import device as D
    W = 8
    shards, cuts = shard_rows(dev_mats, W)
    coll = sharded.DeviceLoopbackCollectives(W, gpu_session.device)
    ctx8 = D.Context(gpu_session.device, gpu_session.lib, W, _lib.FLAG_EMULATE_RANKS, collectives=coll)
    try:
        ctx8.build(shards, to_params([P()] * 5), 20260925, cfg.n_users, cuts[:-1])
        res8 = ctx8.results()
        assert coll.error is None
        _, res = compare_with_oracle_large(gpu_session, mats, [P()] * 5, 20260925, dev_mats=dev_mats, via_context=True,
                                           also=[("8 emulated ranks", [RanksOfAJob(row) for row in res8])])
        work = np.array([sum(int(res8[d][g].stats[0]) for d in range(5)) for g in range(W)], np.float64)
        assert work.max() / work.mean() < 1.05, work           # work-balanced item ranges under the skew
    finally:
        res8 = None
        ctx8.close()
    rows_by_bin = np.sum([st[1:8] for st, _ in res], axis=0)
    assert rows_by_bin[6] > 1000, rows_by_bin                   # the heavy-row class carries real work here


def test_host_level_force_exchange_real_rccl(gpu_session):
    """The JVM-shaped multi-GPU route on a one-GPU box: urcco_cross_occurrence_downsampled with
    options.flags = URCCO_FLAG_FORCE_EXCHANGE -> the process-wide context creates its communicator with ncclCommInitAll and
    the build runs the exchange path (all-reduces, all-gather-v, work-balanced ranges, range-restricted transposition)
    behind the non-gated staging branch.  Config 3 at 1/4 scale, every row against the oracle; then the same call without
    the flag (the default context is keyed on the flags: it must be rebuilt, not silently reused)."""
    import ctypes as C
    import os
    from helpers import check_indicators
    from universal_recommender_amd import _lib, synth
    lib = _lib.load(_lib.DEFAULT_PATH)
    cfg = synth.config3(0.25)
    data = synth.generate(cfg)
    mats = [O.Csr(cfg.n_users, nc, rp, ci) for (_, nc, rp, ci) in data]
    n = len(mats)
    threads = min(os.cpu_count() or 1, O.lib().orc_max_threads())
    ref = O.cross_occurrence_downsampled(mats, [P(), P(), P()], 31, 0, threads)
    arr = (_lib.Dataset * n)()
    for d, m in enumerate(mats):
        arr[d].matrix.n_rows, arr[d].matrix.n_cols = m.n_rows, m.n_cols
        arr[d].matrix.row_ptr, arr[d].matrix.col_idx = m.row_ptr.ctypes.data, m.col_idx.ctypes.data
        arr[d].max_elements_per_row, arr[d].max_interesting_elements = 500, 50
    for flags in (_lib.FLAG_FORCE_EXCHANGE, 0, _lib.FLAG_FORCE_EXCHANGE | _lib.FLAG_UNORDERED_ROWS):
        opts = _lib.Options(device=0, row_rate_mode=0, n_gpus=1, flags=flags)
        out = (_lib.Indicators * n)()
        stats = (_lib.DatasetStats * n)()
        _lib.check(lib.urcco_cross_occurrence_downsampled(arr, n, 31, C.byref(opts), out, stats), lib)
        for d, r in enumerate(ref):
            o = out[d]
            nnz = int(o.nnz)
            got = (np.ctypeslib.as_array(o.row_ptr, shape=(o.n_rows + 1,)).copy(), np.ctypeslib.as_array(o.col_idx, shape=(max(nnz, 1),))[:nnz].copy(),
                   np.ctypeslib.as_array(o.llr, shape=(max(nnz, 1),))[:nnz].copy())
            if flags & _lib.FLAG_UNORDERED_ROWS:
                from helpers import sort_rows
                got = sort_rows(got)
            check_indicators(got, r)
            assert stats[d].pairs == r.pairs and stats[d].nnz_out == nnz
        lib.urcco_free_indicators(out, n)
    assert lib.urcco_shutdown() == 0


def test_host_level_c_abi_config3_size(gpu_session):
    """The host-level entry point a JNI shim binds (urcco_cross_occurrence_downsampled: pageable host CSR in through the
    pinned staging ring, indicator CSR out in pinned host memory) on the FULL config-3 input: every row against the
    oracle, twice (the second call runs on the warm persistent context), then urcco_shutdown."""
    import ctypes as C
    import os
    from helpers import check_indicators
    from universal_recommender_amd import _lib, synth
    lib = _lib.load(_lib.DEFAULT_PATH)
    cfg = synth.config3(1.0)
    data = synth.generate(cfg)
    mats = [O.Csr(cfg.n_users, nc, rp, ci) for (_, nc, rp, ci) in data]
    n = len(mats)
    threads = min(os.cpu_count() or 1, O.lib().orc_max_threads())
    ref = O.cross_occurrence_downsampled(mats, [P(), P(), P()], 77, 0, threads)
    arr = (_lib.Dataset * n)()
    for d, m in enumerate(mats):
        arr[d].matrix.n_rows, arr[d].matrix.n_cols = m.n_rows, m.n_cols
        arr[d].matrix.row_ptr, arr[d].matrix.col_idx = m.row_ptr.ctypes.data, m.col_idx.ctypes.data
        arr[d].max_elements_per_row, arr[d].max_interesting_elements = 500, 50
    opts = _lib.Options(device=0, row_rate_mode=0, n_gpus=1)
    for _ in range(2):
        out = (_lib.Indicators * n)()
        stats = (_lib.DatasetStats * n)()
        _lib.check(lib.urcco_cross_occurrence_downsampled(arr, n, 77, C.byref(opts), out, stats), lib)
        for d, r in enumerate(ref):
            o = out[d]
            nnz = int(o.nnz)
            got = (np.ctypeslib.as_array(o.row_ptr, shape=(o.n_rows + 1,)).copy(), np.ctypeslib.as_array(o.col_idx, shape=(max(nnz, 1),))[:nnz].copy(),
                   np.ctypeslib.as_array(o.llr, shape=(max(nnz, 1),))[:nnz].copy())
            check_indicators(got, r)
            assert stats[d].pairs == r.pairs and stats[d].nnz_out == nnz and stats[d].nnz_raw == mats[d].nnz
        lib.urcco_free_indicators(out, n)
    assert lib.urcco_shutdown() == 0


def test_ingest_path_model_equals_oracle_on_its_own_ids_at_config3_size(gpu_session):
    """Events -> device Preparator -> CCO build at the FULL config-3 size, checked on the ids the device Preparator assigned:
    first-appearance ids permute the (seed, row, col) key of the down-sampling RNG, so this model is NOT the one of the
    generator's ids (90 665 276 vs 90 668 283 pairs) -- the matrices the Preparator built are copied to the host, the oracle
    runs on exactly those, and every down-sampled entry and every indicator row must agree; the Preparator's matrices
    themselves must be the generator's up to the two id permutations (same shapes, same entry count per event type, row-length
    and column-count multisets equal)."""
    from universal_recommender_amd import ingest, synth
    from universal_recommender_amd.device import DevCsr
    cfg = synth.config3(1.0)
    data = synth.generate(cfg)
    rng = np.random.default_rng(1)
    dev = gpu_session.device
    actions = []
    for (name, n_cols, rp, ci) in data:                                  # the stream bench.py's ingest_to_model leg builds
        rows = np.repeat(np.arange(cfg.n_users, dtype=np.int64), np.diff(rp))
        cols = ci.astype(np.int64)
        dup = rng.integers(0, rows.size, rows.size // 10)
        rows, cols = np.concatenate([rows, rows[dup]]), np.concatenate([cols, cols[dup]])
        order = rng.permutation(rows.size)
        uk = rows[order] * np.int64(0x9E3779B97F4A7C15 - (1 << 64)) + 11
        ik = cols[order] * np.int64(0xC2B2AE3D27D4EB4F - (1 << 64)) + 5
        actions.append((name, torch.from_numpy(uk).to(dev), torch.from_numpy(ik).to(dev)))
    dp = ingest.prepare_device(gpu_session, actions, 1)
    gpu_session.synchronize()
    dev_mats, mats = [], []
    for ev, (name, n_cols, rp, ci) in zip(dp.events, data):
        m = ev.matrix
        h = O.Csr(m.n_rows, m.n_cols, m.row_ptr.cpu().numpy(), m.col_idx[: m.nnz_bound].cpu().numpy())
        assert h.nnz == int(rp[-1]) and h.n_rows <= cfg.n_users and h.n_cols <= n_cols
        if name == data[0][0]:                                           # every user has a primary event: the dictionary holds them all
            assert h.n_rows == cfg.n_users
            assert np.array_equal(np.sort(np.diff(h.row_ptr)), np.sort(np.diff(rp)))
        assert np.array_equal(np.sort(np.bincount(h.col_idx, minlength=h.n_cols)), np.sort(np.bincount(ci, minlength=n_cols))[-h.n_cols:])
        dev_mats.append(DevCsr(m.n_rows, m.n_cols, m.row_ptr, m.col_idx, h.nnz))
        mats.append(h)
    _, res = compare_with_oracle_large(gpu_session, mats, [P(), P(), P()], 20260925, dev_mats=dev_mats)
    assert sum(int(st[0]) for st, _ in res) > 80_000_000
