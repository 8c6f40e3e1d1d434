import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def sim_lib():
    """The kernel sources compiled against the TEST-ONLY host simulator (logic checks on CPU; never a parity claim)."""
    from hostsim import build_sim
    from universal_recommender_amd import _lib
    return _lib.load(build_sim.build())


@pytest.fixture(scope="session")
def sim_session(sim_lib):
    import torch
    from universal_recommender_amd.device import DeviceSession
    s = DeviceSession(torch.device("cpu"), sim_lib)
    yield s
    s.close()


@pytest.fixture(scope="session")
def gpu_session():
    """The product path: in-tree liburcco.so (hipcc, gfx950) on cuda:0.  Fails loudly if missing."""
    import torch
    from universal_recommender_amd import _lib
    from universal_recommender_amd.device import DeviceSession
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    import __graft_entry__
    __graft_entry__.build_hip()   # no-op when the in-tree liburcco.so is newer than its sources
    s = DeviceSession(torch.device("cuda", 0), _lib.load(_lib.DEFAULT_PATH))
    yield s
    s.close()
