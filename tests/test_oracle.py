"""The oracle against itself and against independent cross-checks: the C restatement == the pure-Python restatement
(bit-exact ids, identical fp64 LLR: both call libm log in the same expression order), scipy.sparse for the raw
cooccurrence counts, and the decision register's switchable behaviours (D9 row-rate modes, D12 minLLR, D5 zero drop)."""
import numpy as np
import pytest
import scipy.sparse as sp

from helpers import rand_csr
from oracle import c_oracle as O
from oracle import cco_oracle as PO


def _py_rows(m: O.Csr):
    return [m.col_idx[m.row_ptr[r]:m.row_ptr[r + 1]].tolist() for r in range(m.n_rows)]


def _py_ids(m: O.Csr):
    return PO.IndexedDataset(_py_rows(m), PO.BiDictionary([f"u{i}" for i in range(m.n_rows)]), PO.BiDictionary([f"i{j}" for j in range(m.n_cols)]))


@pytest.mark.parametrize("mode", [0, 1, O.RNG_MIX32, 1 | O.RNG_MIX32])
@pytest.mark.parametrize("seed", [0, 0xdeadbeef, -7])
def test_c_oracle_equals_python_oracle(mode, seed):
    rng = np.random.default_rng(abs(seed) % 1000 + mode)
    mats = [rand_csr(rng, 120, 40, 7, zipf_s=1.3), rand_csr(rng, 120, 25, 10), rand_csr(rng, 120, 6, 2, empty_frac=0.4)]
    ps = [(8, 5, None), (12, 4, 0.3), (500, 50, None)]
    ref = PO.cross_occurrence_downsampled([PO.DownsamplableCrossOccurrenceDataset(_py_ids(m), *p) for m, p in zip(mats, ps)], seed, mode)
    got = O.cross_occurrence_downsampled(mats, [O.DatasetParams(*p) for p in ps], seed, mode)
    for g, r in zip(got, ref):
        assert g.pairs == r.pairs
        for i in range(mats[0].n_cols):
            assert g.row(i) == r.rows[i]


def test_rng_is_the_same_stream_in_both_restatements():
    rng = np.random.default_rng(5)
    for s, r, c in rng.integers(0, 2**32, (2000, 3)):
        assert O.lib().orc_u01(int(s), int(r), int(c)) == PO.u01(int(s), int(r), int(c))
    u = np.array([PO.u01(1, r, 7) for r in range(20000)])
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.01


def np_mix32(seed, row, col):
    """cco_oracle.mix32 vectorised (uint32 arithmetic wraps like the C)."""
    with np.errstate(over="ignore"):
        row, col = np.asarray(row, np.uint32), np.asarray(col, np.uint32)
        x = col ^ (row * np.uint32(0x9E3779B1) + np.uint32((int(seed) & 0xFFFFFFFF) * 0x85EBCA77 + 0xC2B2AE3D & 0xFFFFFFFF))
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x7FEB352D)
        x ^= x >> np.uint32(15)
        x *= np.uint32(0x846CA68B)
        x ^= x >> np.uint32(16)
    return x


def test_mix32_is_the_same_stream_in_both_restatements():
    """The 32-bit form of the down-sampling RNG (D10 b, RNG_MIX32): C == Python == the vectorised form the statistics below use."""
    rng = np.random.default_rng(6)
    trip = rng.integers(0, 2**32, (2000, 3))
    L = O.lib()
    for s, r, c in trip:
        assert L.orc_mix32(int(s), int(r), int(c)) == PO.mix32(int(s), int(r), int(c))
        assert L.orc_u01_mix32(int(s), int(r), int(c)) == PO.u01_mix32(int(s), int(r), int(c)) == PO.mix32(int(s), int(r), int(c)) / 4294967296.0
    for s in (0, 1, 0xdeadbeef):
        assert np.array_equal(np_mix32(s, trip[:, 1], trip[:, 2]), np.array([PO.mix32(s, int(r), int(c)) for _, r, c in trip], np.uint32))


def test_mix32_statistics_at_config4_scale():
    """the cheap RNG must sample like a uniform.  On the (row, column) domain of BASELINE config 4 -- 10M users x 2M
    items -- with the keep rule `u01 <= rate`:  per-column keep rates inside binomial bounds for hot columns (rates 500 / count from
    2e-5 to 0.9, 200K draws each), uniform bytes (chi-square), avalanche (one flipped input bit flips ~16 output bits), and independence
    across seeds and across neighbouring rows / columns (correlation of keep decisions)."""
    rng = np.random.default_rng(41)
    n = 200_000
    worst = 0.0
    for col, rate in [(1_234_567, 500 / 23_000_000), (17, 500 / 1_000_000), (1_999_999, 500 / 40_000), (65_536, 0.11), (424_242, 0.5), (3, 0.9)]:
        rows = rng.choice(10_000_000, n, replace=False)          # the users that hold this hot column
        for seed in (1, 20260925):
            kept = int((np_mix32(seed, rows, col).astype(np.float64) / 4294967296.0 <= rate).sum())
            z = (kept - n * rate) / np.sqrt(n * rate * (1 - rate))
            worst = max(worst, abs(z))
    assert worst < 4.5, worst                                      # 12 draws: |z| < 4.5 fails a true uniform once in ~10^5 runs
    # uniformity of every output byte over a dense block of the domain (consecutive rows x consecutive columns: the worst case for a weak mixer)
    r, c = np.meshgrid(np.arange(5_000_000, 5_002_048, dtype=np.uint32), np.arange(1_000_000, 1_001_024, dtype=np.uint32), indexing="ij")
    h = np_mix32(7, r.ravel(), c.ravel())
    for byte in range(4):
        cnt = np.bincount((h >> np.uint32(8 * byte)) & np.uint32(255), minlength=256).astype(np.float64)
        chi2 = ((cnt - h.size / 256) ** 2 / (h.size / 256)).sum()
        assert chi2 < 255 + 5 * np.sqrt(2 * 255), (byte, chi2)     # chi-square with 255 degrees of freedom: mean 255, sd 22.6
    # avalanche: flipping one bit of the row, the column or the seed flips half of the output bits on average
    rows, cols = rng.integers(0, 10_000_000, 20000).astype(np.uint32), rng.integers(0, 2_000_000, 20000).astype(np.uint32)
    base = np_mix32(99, rows, cols)
    pop = lambda x: np.unpackbits(x.view(np.uint8)).sum() / x.size
    for bit in range(0, 24, 3):
        assert 14.5 < pop(base ^ np_mix32(99, rows ^ np.uint32(1 << bit), cols)) < 17.5, bit
    for bit in range(0, 21, 3):
        assert 14.5 < pop(base ^ np_mix32(99, rows, cols ^ np.uint32(1 << bit))) < 17.5, bit
    for bit in range(0, 32, 5):
        assert 14.5 < pop(base ^ np_mix32(99 ^ (1 << bit), rows, cols)) < 17.5, bit
    # independence: keep decisions of two seeds, of a row and its neighbour, of a column and its neighbour are uncorrelated
    m = 2_000_000
    rows, cols = rng.integers(0, 10_000_000, m).astype(np.uint32), rng.integers(0, 2_000_000, m).astype(np.uint32)
    k = lambda s, rr, cc: (np_mix32(s, rr, cc) < np.uint32(0x40000000)).astype(np.float64)   # rate 1/4
    a = k(5, rows, cols)
    for other in (k(6, rows, cols), k(5, rows + np.uint32(1), cols), k(5, rows, cols + np.uint32(1)), k(5, rows ^ np.uint32(0x10000), cols)):
        corr = np.corrcoef(a, other)[0, 1]
        assert abs(corr) < 5 / np.sqrt(m), corr


def test_counts_against_scipy():
    rng = np.random.default_rng(9)
    a, b = rand_csr(rng, 500, 60, 8), rand_csr(rng, 500, 90, 12)
    A = sp.csr_matrix((np.ones(a.nnz, np.int64), a.col_idx, a.row_ptr), shape=(a.n_rows, a.n_cols))
    B = sp.csr_matrix((np.ones(b.nnz, np.int64), b.col_idx, b.row_ptr), shape=(b.n_rows, b.n_cols))
    K = (A.T @ B).toarray()
    counts = PO.at_b(_py_rows(a), _py_rows(b), a.n_cols)
    for i in range(a.n_cols):
        assert {j: int(K[i, j]) for j in np.nonzero(K[i])[0]} == counts[i]
    assert int(K.sum()) == PO.count_pairs(_py_rows(a), _py_rows(b))
    assert np.array_equal(np.asarray(A.sum(axis=0)).ravel(), O.column_counts(a))


def test_llr_known_answers_and_properties():
    assert O.lib().orc_llr_k(1, 1, 0, 2) == pytest.approx(1.7260924347106847, abs=1e-15)
    assert O.lib().orc_llr_k(1, 1, 1, 1) == 0.0
    rng = np.random.default_rng(2)
    for _ in range(2000):
        k = [int(x) for x in rng.integers(0, 1000, 4)]
        v = O.lib().orc_llr_k(*k)
        assert v >= 0.0 and v == PO.log_likelihood_ratio(*k)
        assert abs(v - O.lib().orc_llr_k(k[0], k[2], k[1], k[3])) <= 1e-9 * max(v, 1.0)   # transposing the table


def test_row_rate_modes_d9():
    """Mahout's Int/Int row rate drops a row longer than the cap entirely; the fractional switch thins it."""
    m = O.Csr.from_rows([list(range(40)), [1, 2, 3]], 40)
    raw = O.column_counts(m)
    intdiv = O.downsample(m, raw, 3, 10, 0)
    frac = O.downsample(m, raw, 3, 10, 1)
    assert np.diff(intdiv.row_ptr).tolist() == [0, 3]
    assert 0 < np.diff(frac.row_ptr)[0] < 40 and np.diff(frac.row_ptr)[1] == 3


def test_min_llr_and_zero_drop():
    rows = [[0, 1], [0, 1], [0], [1]]
    ids = PO.IndexedDataset(rows, PO.BiDictionary("abcd"), PO.BiDictionary(["x", "y"]))
    full = PO.cross_occurrence_downsampled([PO.DownsamplableCrossOccurrenceDataset(ids)], 1)[0]
    assert all(s > 0 for r in full.rows for _, s in r)                       # D5: zeros never materialise
    hi = PO.cross_occurrence_downsampled([PO.DownsamplableCrossOccurrenceDataset(ids, 500, 50, 1e9)], 1)[0]
    assert all(len(r) == 0 for r in hi.rows)                                 # D12: minLLR filters before the cut
    assert PO.seed_to_int(0xdeadbeef) == -559038737 and PO.seed_to_int(2**40 + 5) == 5   # D14


def test_parallel_c_oracle_paths_against_scipy():
    """The OpenMP forms of orc_downsample / orc_transpose (matrices >= 2^22 entries take the threaded transposition)
    equal the sequential definitions: transposition vs scipy's CSC (rows ascending inside a column), down-sampling vs
    a vectorised numpy evaluation of the same keep rule."""
    rng = np.random.default_rng(31)
    m = rand_csr(rng, 150_000, 50_000, 34, zipf_s=1.0)
    assert m.nnz >= (1 << 22)
    cp, ri = O.transpose(m)
    M = sp.csr_matrix((np.ones(m.nnz, np.int8), m.col_idx, m.row_ptr), shape=(m.n_rows, m.n_cols)).tocsc()
    M.sort_indices()
    assert np.array_equal(cp, M.indptr) and np.array_equal(ri, M.indices)
    raw = O.column_counts(m)
    ds = O.downsample(m, raw, 77, 40, 1)
    L = O.lib()
    rows = np.repeat(np.arange(m.n_rows), np.diff(m.row_ptr))
    n_row = np.diff(m.row_ptr)[rows].astype(np.float64)
    rate = np.minimum(np.minimum(n_row, 40.0) / n_row, np.minimum(raw[m.col_idx], 40.0) / raw[m.col_idx])
    pick = rng.integers(0, m.nnz, 20000)
    u = np.array([L.orc_u01(77, int(rows[e]), int(m.col_idx[e])) for e in pick])
    keep = u <= rate[pick]
    pos = np.searchsorted(ds.row_ptr, np.arange(ds.nnz), side="right") - 1
    kept = set(zip(pos.tolist(), ds.col_idx.tolist()))
    assert all(((int(rows[e]), int(m.col_idx[e])) in kept) == bool(k) for e, k in zip(pick, keep))
    assert np.all(np.diff(ds.row_ptr) <= np.diff(m.row_ptr))
