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


@pytest.mark.parametrize("mode", [0, 1])
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
