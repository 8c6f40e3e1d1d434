"""Round-2 CANDIDATE code paths (off by default, selected by urcco_session_set_debug bits that do not change results):
their logic is checked here on the host simulator so that they are ready to be measured on hardware."""
import numpy as np
import pytest

import test_sim_kernel_logic as logic


@pytest.fixture
def with_debug(sim_session):
    def run(bits, fn, *args):
        sim_session.set_debug(bits)
        try:
            return fn(sim_session, *args)
        finally:
            sim_session.set_debug(0)
    return run


@pytest.mark.parametrize("case", [logic.test_small_three_events_all_modes, logic.test_hash_tables_and_all_bins,
                                  logic.test_all_equal_llr_ties_cut_by_column, logic.test_item_range_slices_concatenate],
                         ids=lambda f: f.__name__)
def test_bitonic_rank_of_survivors(with_debug, case):
    with_debug(1024, case)


@pytest.mark.parametrize("case", [logic.test_partitioned_column_counts_large_matrix, logic.test_large_matrix_full_pipeline],
                         ids=lambda f: f.__name__)
def test_lds_staged_bucket_scatter(with_debug, case):
    with_debug(2048, case)


@pytest.mark.parametrize("mode", [0, 1])
def test_tile_table_carries_the_row_start_bit(with_debug, mode):
    with_debug(4096, logic.test_row_scan_tile_edges, mode)
    with_debug(4096, logic.test_downsample_row_base_matches_sharded_rows)


@pytest.mark.parametrize("case", [logic.test_small_three_events_all_modes, logic.test_hash_tables_and_all_bins,
                                  logic.test_packed_count_overflow_goes_global, logic.test_large_matrix_full_pipeline],
                         ids=lambda f: f.__name__)
def test_claim_first_insert(with_debug, case):
    with_debug(8192, case)
