"""The N > 1 path on CPU: world_size-2 and -3 `gloo` process groups drive the library's multi-rank build
(urcco_context_build_device: user-range shards, count all-reduces, all-gather-v of the down-sampled shards, all-to-all-v of the
primary's CSC fragments, work-balanced item ranges) end to end, the collectives going through the urcco_collectives callbacks instead of RCCL.  Compute underneath is the
kernel sources on the TEST-ONLY host simulator; the result must equal the single-process oracle exactly the way the
single-GPU path does -- i.e. the sharding is invisible."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, uneven, q):
    try:
        for p in (ROOT, HERE):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ["OMP_NUM_THREADS"] = "2"
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        from helpers import check_indicators, rand_csr, to_dev, to_params
        from hostsim import build_sim
        from oracle import c_oracle as O
        from universal_recommender_amd import _lib, sharded
        ctx = sharded.make_context(torch.device("cpu"), _lib.load(build_sim.build()))
        rng = np.random.default_rng(77)                       # same matrices on every rank
        if uneven == "skew5":                                 # five event types with config 5's skew (hot head + heavy users)
            from universal_recommender_amd import synth
            cfg = synth.config5(0.0003)
            n_users = cfg.n_users
            mats = [O.Csr(n_users, nc, rp, ci) for (_, nc, rp, ci) in synth.generate(cfg)]
            params = [O.DatasetParams(60, 20, None), O.DatasetParams(80, 20, None), O.DatasetParams(500, 10, None), O.DatasetParams(40, 50, 0.5),
                      O.DatasetParams(500, 50, None)]
            assert len(mats) == 5 and max(int(np.diff(m.row_ptr).max()) for m in mats) > 80
        else:
            n_users = 1201
            mats = [rand_csr(rng, n_users, 300, 9, zipf_s=1.2), rand_csr(rng, n_users, 700, 14), rand_csr(rng, n_users, 11, 2, empty_frac=0.3)]
            params = [O.DatasetParams(30, 10, None), O.DatasetParams(40, 12, None), O.DatasetParams(500, 50, 0.1)]
        if uneven == "empty":                                 # rank 0 holds no users at all
            cuts = [0, 0] + [n_users * r // (world - 1) for r in range(1, world)]
        elif uneven and world == 2:
            cuts = [0, 900, n_users]
        else:
            cuts = [n_users * r // world for r in range(world + 1)]
        lo, hi = cuts[rank], cuts[rank + 1]
        if uneven == "rebuild":
            # three builds on ONE context, the middle one smaller and with other cuts and another seed: every buffer of the exchange
            # (masks, masked lengths, packed rows, fragments) is reused at a different size and must not leak anything from the build before
            small = [O.Csr(400, m.n_cols, m.row_ptr[:401].copy(), m.col_idx[:m.row_ptr[400]].copy()) for m in mats]
            for ms, seed, cs in ((mats, 5, cuts), (small, 6, [0, 150, 400] if world == 2 else [400 * r // world for r in range(world + 1)]), (mats, 7, cuts)):
                l2, h2 = cs[rank], cs[rank + 1]
                sh2 = [O.Csr(h2 - l2, m.n_cols, m.row_ptr[l2:h2 + 1] - m.row_ptr[l2], m.col_idx[m.row_ptr[l2]:m.row_ptr[h2]]) for m in ms]
                r2 = sharded.cross_occurrence_sharded(ctx, [to_dev(s, "cpu") for s in sh2], to_params(params), seed, ms[0].n_rows, l2)
                f2 = sharded.gather_indicators_to_host(r2)
                ref2 = O.cross_occurrence_downsampled(ms, params, seed)
                for got, r in zip(f2, ref2):
                    check_indicators(got, r, exact_ids=True)
        shards = [O.Csr(hi - lo, m.n_cols, m.row_ptr[lo:hi + 1] - m.row_ptr[lo], m.col_idx[m.row_ptr[lo]:m.row_ptr[hi]]) for m in mats]
        res = sharded.cross_occurrence_sharded(ctx, [to_dev(s, "cpu") for s in shards], to_params(params), 2024, n_users, lo)
        full = sharded.gather_indicators_to_host(res)
        ranges = sharded.gather_item_ranges(res, mats[0].n_cols)
        ref = O.cross_occurrence_downsampled(mats, params, 2024)
        pairs = [int(i.stats[0]) for i in res.indicators]
        all_pairs = [None] * world
        dist.all_gather_object(all_pairs, pairs)
        for d, (got, r) in enumerate(zip(full, ref)):
            check_indicators(got, r, exact_ids=uneven != "skew5")
            assert sum(p[d] for p in all_pairs) == r.pairs
        assert ranges[0] == 0 and ranges[-1] == mats[0].n_cols and len(ranges) == world + 1
        q.put((rank, "ok", ranges))
        ctx.close()
        dist.destroy_process_group()
    except Exception as e:  # surface the failure in the parent
        import traceback
        q.put((rank, "fail: " + traceback.format_exc(), None))


@pytest.mark.parametrize("world,uneven", [(2, False), (2, True), (3, False), (4, False), (2, "empty"), (2, "skew5"), (2, "rebuild"), (3, "rebuild")])
def test_sharded_equals_single_process_oracle(world, uneven, sim_lib):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, uneven, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, status, _ in results:
        assert status == "ok", f"rank {rank}: {status}"
    ranges = [r[2] for r in results]
    assert all(r == ranges[0] for r in ranges), "ranks disagree on the item ranges"
