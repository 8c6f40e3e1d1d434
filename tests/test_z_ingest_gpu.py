"""PARITY (-m gpu) of the device-side Preparator: the ingest kernels on a real MI355X through the C ABI against the
oracle's Preparator.prepare -- dictionaries and matrices bit for bit (the cases of tests/test_sim_ingest.py on
hardware, plus one stream at a size where every kernel runs many blocks)."""
import numpy as np
import pytest

import test_sim_ingest as logic

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("min_events", [None, 2])
def test_prepare_matches_the_oracle_on_gpu(gpu_session, min_events):
    logic.test_prepare_matches_the_oracle(gpu_session, min_events)


def test_prepare_degenerate_and_long_rows_on_gpu(gpu_session):
    logic.test_prepare_handmade_and_degenerate_streams(gpu_session)
    logic.test_prepare_long_rows_take_the_block_sorts(gpu_session)


def test_prepare_larger_stream_on_gpu(gpu_session):
    rng = np.random.default_rng(99)
    acts = logic.random_actions(rng, 20000, [8000, 30000], [150000, 250000], n_types=2, user_pool_extra=3000)
    logic.check_prepare(gpu_session, acts, 2)


def test_preparator_mirror_device_equals_host_on_gpu(gpu_session):
    logic.test_preparator_mirror_device_equals_host(gpu_session)


def test_events_to_model_without_leaving_the_device_on_gpu(gpu_session):
    logic.test_events_to_model_without_leaving_the_device(gpu_session, gpu_session.lib)


def test_native_hash_and_collision_check_on_gpu(gpu_session):
    logic.test_native_string_hash_known_answers_and_collision_check(gpu_session, gpu_session.lib)
