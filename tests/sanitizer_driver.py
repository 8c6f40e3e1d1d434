"""Run under LD_PRELOAD=libasan (tests/test_sanitizers.py): drives the sanitized C ABI (host level incl. error paths, context
reuse, multi-"GPU" orchestration with looped-back collectives, ingest helpers) and the sanitized C oracle with plain
ctypes + numpy -- no torch in this process (the simulator's "device memory" is host memory)."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

sim_path, oracle_path = sys.argv[1], sys.argv[2]
import importlib.util
spec = importlib.util.spec_from_file_location("urcco_lib", os.path.join(ROOT, "universal-recommender_amd", "_lib.py"))
_lib = importlib.util.module_from_spec(spec)
spec.loader.exec_module(_lib)
lib = _lib._bind(sim_path)
orc = C.CDLL(oracle_path)
orc.orc_llr.restype = C.c_double
orc.orc_llr.argtypes = [C.c_int64] * 4
assert abs(orc.orc_llr(2, 1, 1, 4) - 1.7260924347106847) < 1e-12


def rand_csr(rng, n_rows, n_cols, avg):
    deg = rng.poisson(avg, n_rows)
    rows = np.repeat(np.arange(n_rows, dtype=np.int64), deg)
    cols = rng.integers(0, n_cols, rows.size)
    key = np.unique(rows * n_cols + cols)
    r = key // n_cols
    rp = np.zeros(n_rows + 1, np.int64)
    np.cumsum(np.bincount(r, minlength=n_rows), out=rp[1:])
    return rp, (key - r * n_cols).astype(np.int32)


def datasets(mats, max_rows=30, k=10):
    arr = (_lib.Dataset * len(mats))()
    for d, (n_rows, n_cols, rp, ci) in enumerate(mats):
        arr[d].matrix.n_rows, arr[d].matrix.n_cols = n_rows, n_cols
        arr[d].matrix.row_ptr, arr[d].matrix.col_idx = rp.ctypes.data, ci.ctypes.data
        arr[d].max_elements_per_row, arr[d].max_interesting_elements = max_rows, k
    return arr


rng = np.random.default_rng(1)
mats = [(4000, 900, *rand_csr(rng, 4000, 900, 9)), (4000, 1500, *rand_csr(rng, 4000, 1500, 14)), (4000, 12, *rand_csr(rng, 4000, 12, 2))]
opts = _lib.Options(device=0, row_rate_mode=0, n_gpus=1)
pairs = None
for it in range(3):                                            # process-wide context: created, reused, shut down, re-created
    out = (_lib.Indicators * 3)()
    stats = (_lib.DatasetStats * 3)()
    st = lib.urcco_cross_occurrence_downsampled(datasets(mats), 3, 99, C.byref(opts), out, stats)
    assert st == 0, lib.urcco_last_error()
    p = [int(s.pairs) for s in stats]
    assert pairs is None or p == pairs
    pairs = p
    assert all(int(o.nnz) > 0 for o in out[:2])
    lib.urcco_free_indicators(out, 3)
    if it == 1:
        assert lib.urcco_shutdown() == 0
# error paths: bad column index, non-monotone row_ptr, mismatched rows, non-positive limits
bad_ci = mats[0][3].copy()
bad_ci[5] = 10_000
for bad in ([(4000, 900, mats[0][2], bad_ci)], [(4000, 900, mats[0][2][::-1].copy(), mats[0][3])], [mats[0], (10, 5, np.zeros(11, np.int64), np.zeros(1, np.int32))]):
    out = (_lib.Indicators * len(bad))()
    rp0 = bad[0][2]
    if rp0[0] != 0:
        rp0 = rp0.copy()
    st = lib.urcco_cross_occurrence_downsampled(datasets(bad), len(bad), 1, C.byref(opts), out, None)
    assert st == _lib.BAD_ARG, (st, lib.urcco_last_error())
out = (_lib.Indicators * 1)()
assert lib.urcco_cross_occurrence_downsampled(datasets(mats[:1], 0, 10), 1, 1, C.byref(opts), out, None) == _lib.BAD_ARG
assert lib.urcco_cross_occurrence_downsampled(None, 0, 1, C.byref(opts), out, None) == _lib.BAD_ARG
lib.urcco_shutdown()
# string hashing helper
strs = [f"user-{i}".encode() for i in range(70000)]
off = np.zeros(len(strs) + 1, np.int64)
np.cumsum([len(s) for s in strs], out=off[1:])
blob = np.frombuffer(b"".join(strs), np.uint8)
keys = np.empty(len(strs), np.uint64)
assert lib.urcco_hash_strings(blob.ctypes.data, off.ctypes.data, len(strs), C.c_uint64(0), keys.ctypes.data) == 0
assert np.unique(keys).size == len(strs)
# several simulated GPUs in one process, collectives looped back (orchestration, staging ring, pinned pool, output assembly)
os.environ["HIPSIM_DEVICE_COUNT"] = "3"
pend = {}


def _gs(user):
    return 0


def _ge(user):
    if not pend:
        return 0
    n_ops = len(next(iter(pend.values())))
    for j in range(n_ops):
        ops = {r: v[j] for r, v in pend.items()}
        kind = next(iter(ops.values()))[0]
        if kind == "ar":
            views = [np.ctypeslib.as_array(((C.c_int32 if o[3] == 0 else C.c_int64) * o[2]).from_address(o[1])) for o in ops.values()]
            tot = np.sum(views, axis=0, dtype=views[0].dtype)
            for v in views:
                v[:] = tot
        elif kind == "aa":
            pieces = {(r, q): bytes((C.c_char * o[3][q]).from_address(o[1] + o[2][q])) if o[3][q] > 0 else b"" for r, o in ops.items() for q in range(3)}
            for r, o in ops.items():
                for p in range(3):
                    assert len(pieces[(p, r)]) == o[6][p]
                    if o[6][p] > 0:
                        C.memmove(o[4] + o[5][p], pieces[(p, r)], o[6][p])
        else:
            pieces = {r: bytes((C.c_char * o[4][r]).from_address(o[1])) if o[4][r] > 0 else b"" for r, o in ops.items()}
            for r, o in ops.items():
                for p in range(3):
                    if o[4][p] > 0:
                        C.memmove(o[2] + o[3][p], pieces[p], o[4][p])
    pend.clear()
    return 0


def _ar(user, rank, buf, count, dtype, stream):
    pend.setdefault(rank, []).append(("ar", buf, count, dtype))
    return 0


def _ag(user, rank, send, recv, offs, cnts, stream):
    pend.setdefault(rank, []).append(("ag", send, recv, [offs[r] for r in range(3)], [cnts[r] for r in range(3)]))
    return 0


def _aa(user, rank, send, soff, scnt, recv, roff, rcnt, stream):
    w = range(3)
    pend.setdefault(rank, []).append(("aa", send, [soff[r] for r in w], [scnt[r] for r in w], recv, [roff[r] for r in w], [rcnt[r] for r in w]))
    return 0


cbs = (_lib.GROUP_FN(_gs), _lib.GROUP_FN(_ge), _lib.ALL_REDUCE_FN(_ar), _lib.ALL_GATHER_V_FN(_ag), _lib.ALL_TO_ALL_V_FN(_aa))
coll = _lib.Collectives(None, *cbs)
comm = _lib.CommConfig(world_size=3, first_rank=0)
comm.collectives = C.pointer(coll)
opts3 = _lib.Options(device=0, row_rate_mode=0, n_gpus=3)
ctx = C.c_void_p()
assert lib.urcco_context_create(C.byref(opts3), C.byref(comm), C.byref(ctx)) == 0, lib.urcco_last_error()
out = (_lib.Indicators * 3)()
stats = (_lib.DatasetStats * 3)()
assert lib.urcco_context_cross_occurrence(ctx, datasets(mats), 3, 99, out, stats) == 0, lib.urcco_last_error()
assert [int(s.pairs) for s in stats] == pairs, "3 simulated GPUs form the same pairs as one"
lib.urcco_free_indicators(out, 3)
lib.urcco_context_destroy(ctx)
lib.urcco_shutdown()
print("SANITIZED_RUN_OK")
