"""Property tests (hypothesis) of the kernel logic on the host simulator against the oracle: arbitrary small shapes,
densities, caps, k, minLLR, seeds, row-rate modes and item ranges -- the corners a fixed list of cases misses
(k = 1, k larger than the column count, caps of 1, empty secondary matrices, single-column matrices ...)."""
import numpy as np
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from helpers import compare_with_oracle, rand_csr
from oracle import c_oracle as O


@st.composite
def problems(draw):
    seed = draw(st.integers(0, 2**31 - 1))
    rng = np.random.default_rng(seed)
    n_users = draw(st.integers(1, 400))
    n_ds = draw(st.integers(1, 3))
    mats, params = [], []
    for d in range(n_ds):
        n_cols = draw(st.sampled_from([1, 2, 7, 33, 120, 700, 5000]))
        avg = draw(st.sampled_from([0.3, 1.5, 4.0, 12.0]))
        zipf = draw(st.sampled_from([0.0, 1.0, 1.6]))
        empty = draw(st.sampled_from([0.0, 0.5]))
        mats.append(rand_csr(rng, n_users, n_cols, avg, zipf_s=zipf, empty_frac=empty))
        params.append(O.DatasetParams(draw(st.sampled_from([1, 3, 10, 500])), draw(st.sampled_from([1, 2, 5, 50, 300])),
                                      draw(st.sampled_from([None, None, 0.0, 0.7, 5.0]))))
    mode = draw(st.integers(0, 1))
    run_seed = draw(st.integers(-2**31, 2**31 - 1))
    n_items = mats[0].n_cols
    lo = draw(st.integers(0, n_items))
    hi = draw(st.integers(lo, n_items))
    full = draw(st.booleans())
    return mats, params, run_seed, mode, (0, n_items) if full else (lo, hi)


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large,
                                                                   HealthCheck.function_scoped_fixture])
@given(problems())
def test_random_problems_match_the_oracle(sim_session, problem):
    mats, params, run_seed, mode, (lo, hi) = problem
    compare_with_oracle(sim_session, mats, params, run_seed, mode, lo, hi)


def test_micro_class_unsplit_in_small_builds():
    """Builds below a million item rows keep ONE micro list (what a rank of a sharded build runs): the kernel-logic cases of the micro class once more in a
    subprocess whose library reads the default threshold."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, URCCO_MICRO_SPLIT_ROWS="1000000")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "tests/test_sim_kernel_logic.py", "-k", "micro or small_three or empty_and_ragged"],
                       cwd=root, env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
