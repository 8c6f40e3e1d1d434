"""Helper of tests/test_gpu_parity.py::test_exchange_path_over_rccl_on_one_gpu -- launched under torch.distributed.run
with ONE rank: initialises the `nccl` (= RCCL) backend on cuda:0 and runs the multi-GPU exchange path of sharded.py
(all-reduces, padded all-gather, work-balanced ranges, range-restricted transpose) in a one-rank group, then checks the
result against the oracle.  Real multi-GPU boxes are not available to the tests; this exercises every RCCL call the
N > 1 path makes (dtypes, device tensors, stream ordering) on hardware."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    from helpers import check_indicators, rand_csr, to_dev, to_params
    from oracle import c_oracle as O
    from universal_recommender_amd import _lib, sharded
    from universal_recommender_amd.device import DeviceSession
    sess = DeviceSession(dev, _lib.load(_lib.DEFAULT_PATH))
    rng = np.random.default_rng(31)
    n_users = 20000
    mats = [rand_csr(rng, n_users, 3000, 9, zipf_s=1.1), rand_csr(rng, n_users, 5000, 14), rand_csr(rng, n_users, 30, 2, empty_frac=0.2)]
    params = [O.DatasetParams(60, 20, None), O.DatasetParams(60, 20, None), O.DatasetParams(500, 50, None)]
    res = sharded.cross_occurrence_sharded(sess, [to_dev(m, dev) for m in mats], to_params(params), 99, n_users, 0, force_exchange=True)
    dist.barrier()
    torch.cuda.synchronize()
    ref = O.cross_occurrence_downsampled(mats, params, 99)
    for ind, r in zip(res.indicators, ref):
        assert int(ind.stats[0]) == r.pairs
        check_indicators(ind.to_host(), r)
    assert all(b[0] == 0 and b[-1] == mats[0].n_cols for b in res.item_ranges)
    # the same build with the A'B_d of each event type on its own HIP stream behind its own (asynchronous) gather
    from universal_recommender_amd.device import SessionPool
    pool = SessionPool(dev, len(mats), sess.lib)
    for _ in range(3):
        res2 = sharded.cross_occurrence_sharded(sess, [to_dev(m, dev) for m in mats], to_params(params), 99, n_users, 0, force_exchange=True, pool=pool)
        torch.cuda.synchronize()
        for i1, i2 in zip(res.indicators, res2.indicators):
            n1 = int(i1.row_ptr[-1])
            for t1, t2 in ((i1.row_ptr, i2.row_ptr), (i1.col_idx[:n1], i2.col_idx[:n1]), (i1.llr[:n1], i2.llr[:n1]), (i1.stats, i2.stats)):
                assert torch.equal(t1, t2), "stream-per-event sharded build differs from the one-stream build"
    pool.close()
    dist.destroy_process_group()
    print("EXCHANGE_PATH_OK")


if __name__ == "__main__":
    main()
