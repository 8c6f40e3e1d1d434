"""Helper of tests/test_gpu_parity.py::test_select_overlay_race_negative_control -- run in its own process, because the
round-3 race it re-creates may end in a GPU memory fault.  Two builds of the same workload on cuda:0 with the first wave of
every multi-wave team delayed before it reads the select histograms (debug 131072): the first with the barrier that
round 4 added in front of the ambiguous-set copy-out (cco_rows.hip, `SHARE && T != WAVE`), the second with that barrier
skipped (debug 262144 = the code as round 3 shipped it).  Prints RACE_REPRODUCED when the second build's indicator rows
differ from the first's (or when it faults), RACE_NOT_REPRODUCED when they are identical."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def workload():
    """Rows whose select is CERTAIN to consume the clobbered words.  Two groups of items: X (121 items) held together by
    7 users, Y (100 items) held -- with X -- by 2 of them.  A row of X then has 120 candidates tied at the top LLR (more than
    k = 50, at most the 128 the ambiguous set holds: the select finishes in its FIRST pass, whose histogram the 120 copied
    keys overwrite completely) and 100 weaker ones, 221 distinct columns in all: the team's first wave owns the lowest columns --
    the ones the tie at the cut selects -- and decides about them with whatever it read.  Work 5 x 121 + 2 x 221 = 1047 pairs -> the 256-thread
    small-block class (as are the rows of Y: 442 pairs, 221 columns)."""
    from oracle import c_oracle as O
    n_users, n_items, nx, ny = 2000, 512, 121, 100
    rows = [np.zeros(0, np.int64) for _ in range(n_users)]
    u = 0
    for b in range(2):
        base = b * (nx + ny)
        for _ in range(5):
            rows[u] = np.arange(base, base + nx, dtype=np.int64)
            u += 1
        for _ in range(2):
            rows[u] = np.arange(base, base + nx + ny, dtype=np.int64)
            u += 1
    rp = np.zeros(n_users + 1, np.int64)
    np.cumsum([len(r) for r in rows], out=rp[1:])
    a = O.Csr(n_users, n_items, rp, np.concatenate(rows).astype(np.int32))
    return [a, a]


N_ROWS_SMALL_BLOCK = 2 * (121 + 100)


def main():
    try:
        import resource
        resource.setrlimit(resource.RLIMIT_CORE, (0, 0))
    except Exception:
        pass
    from helpers import run_device
    from oracle import c_oracle as O
    from universal_recommender_amd import _lib
    from universal_recommender_amd.device import DeviceSession
    sess = DeviceSession(torch.device("cuda", 0), _lib.load(_lib.DEFAULT_PATH))
    mats = workload()
    params = [O.DatasetParams(500, 50, None)] * len(mats)
    sess.set_debug(131072)
    good = [o.to_host() for o in run_device(sess, mats, params, 77)]
    sess.set_debug(131072 | 262144)
    try:
        bad = [o.to_host() for o in run_device(sess, mats, params, 77)]
    except Exception as e:  # a HIP error: the clobbered threshold let a garbage column through
        print("RACE_REPRODUCED (fault):", e, flush=True)
        os._exit(0)
    rows = 0
    for (rp, ci, llr), (rp2, ci2, llr2) in zip(good, bad):
        n = np.diff(rp) != np.diff(rp2)
        rows += int(n.sum())
        if not n.any() and not (np.array_equal(ci, ci2) and np.array_equal(llr, llr2)):
            rows += 1
    print(("RACE_REPRODUCED" if rows else "RACE_NOT_REPRODUCED") + f" ({rows} indicator rows differ)", flush=True)
    os._exit(0)  # no teardown on a context that may have faulted


if __name__ == "__main__":
    main()
