import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


# The micro class's sub-lists (rows of <= 16 / <= 32 pairs share a wave four / two at a time) are only built for builds of a million item rows and
# more (cco_rows.hip, micro_split_for): the tests' matrices are small, so the threshold (read once by the library) is lowered here -- every test
# runs the three sub-list kernels; tests/test_sim_properties.py runs the unsplit form in a subprocess.
os.environ.setdefault("URCCO_MICRO_SPLIT_ROWS", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    try:  # a GPU fault aborts the process; with ~100 GB mapped the core dump alone took ten minutes on a gpurun box
        import resource
        resource.setrlimit(resource.RLIMIT_CORE, (0, 0))
    except Exception:
        pass


@pytest.fixture(scope="session")
def sim_lib():
    """The kernel sources compiled against the TEST-ONLY host simulator (logic checks on CPU; never a parity claim)."""
    from hostsim import build_sim
    from universal_recommender_amd import _lib
    if os.environ.get("HIPSIM_VARIANT") == "bounds":      # tests/test_sim_guard.py: LDS array indices checked
        return _lib.load(build_sim.build_bounds())
    return _lib.load(build_sim.build())


def guarded_tensor(lib, n, dtype):
    """HIPSIM_GUARD runs: a CPU tensor whose storage ends at a PROT_NONE page of the simulator (tests/hostsim/hipsim.cpp), so a
    kernel that runs past a caller-owned input or output faults."""
    import ctypes as C
    import weakref
    import numpy as np
    import torch
    item = torch.empty(0, dtype=dtype).element_size()
    lib.hipsim_guard_malloc.restype = C.c_void_p
    lib.hipsim_guard_malloc.argtypes = [C.c_size_t]
    lib.hipsim_guard_release.argtypes = [C.c_void_p]
    nbytes = max(int(n), 1) * item
    mode = os.environ.get("HIPSIM_GUARD")
    body = nbytes if mode == "3" else ((nbytes + 15) & ~15 if mode != "2" else (nbytes + 3) & ~3)
    ptr = lib.hipsim_guard_malloc(nbytes)
    buf = (C.c_uint8 * nbytes).from_address(ptr + (body - nbytes))  # the requested bytes END where the allocation ends (mode 3: START behind a guard page)
    arr = np.frombuffer(buf, dtype=np.uint8)
    t = torch.from_numpy(arr).view(dtype)[: int(n)]
    weakref.finalize(arr, lib.hipsim_guard_release, ptr)
    return t


@pytest.fixture(scope="session")
def sim_session(sim_lib):
    import torch
    from universal_recommender_amd.device import DeviceSession
    s = DeviceSession(torch.device("cpu"), sim_lib)
    if os.environ.get("HIPSIM_GUARD"):
        import helpers
        s.empty = lambda n, dtype: guarded_tensor(sim_lib, n, dtype)
        helpers.GUARD_LIB = sim_lib
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
