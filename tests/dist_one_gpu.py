"""Helper of tests/test_gpu_parity.py::test_exchange_path_over_rccl_on_one_gpu -- launched under torch.distributed.run
with ONE rank: runs the library's multi-GPU exchange path (RCCL all-reduces, all-gather-v by grouped send / recv,
work-balanced ranges, range-restricted transpose -- csrc/urcco_context.hip) in a one-rank communicator on cuda:0, then
checks the result against the oracle.  Real multi-GPU boxes are not available to the tests; this exercises every RCCL call the
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
    lib = _lib.load(_lib.DEFAULT_PATH)
    rng = np.random.default_rng(31)
    n_users = 20000
    mats = [rand_csr(rng, n_users, 3000, 9, zipf_s=1.1), rand_csr(rng, n_users, 5000, 14), rand_csr(rng, n_users, 30, 2, empty_frac=0.2)]
    params = [O.DatasetParams(60, 20, None), O.DatasetParams(60, 20, None), O.DatasetParams(500, 50, None)]
    ref = O.cross_occurrence_downsampled(mats, params, 99)
    first = None
    for flags in (_lib.FLAG_FORCE_EXCHANGE | _lib.FLAG_SINGLE_STREAM, _lib.FLAG_FORCE_EXCHANGE):   # one stream, then a stream per event type
        ctx = sharded.make_context(dev, lib, flags=flags)     # ncclGetUniqueId + ncclCommInitRank inside the library
        for _ in range(3):
            res = sharded.cross_occurrence_sharded(ctx, [to_dev(m, dev) for m in mats], to_params(params), 99, n_users, 0)
            host = [ind.to_host() for ind in res.indicators]
            for ind, h, r in zip(res.indicators, host, ref):
                assert int(ind.stats[0]) == r.pairs
                check_indicators(h, r)
                assert ind.item_lo == 0 and ind.item_hi == mats[0].n_cols
            if first is None:
                first = host
            for h1, h2 in zip(first, host):
                for x, y in zip(h1, h2):
                    assert np.array_equal(x, y), "stream-per-event sharded build differs from the one-stream build"
        ctx.close()
    dist.barrier()
    dist.destroy_process_group()
    print("EXCHANGE_PATH_OK")


if __name__ == "__main__":
    main()
