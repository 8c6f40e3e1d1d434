"""Buffer overruns: the simulator-backed kernel and context tests once more with every library allocation (hipMalloc, each
sub-buffer of a session's scratch arena, the context's device buffers without their growth slack) and every test-side input /
output tensor ENDING at a PROT_NONE page (tests/hostsim/hipsim.cpp: HIPSIM_GUARD).  A kernel that reads or writes past the end
of a buffer -- a vector load over the tail, row_ptr[n_rows + 1], an upper-bound launch that forgets its live length -- dies
with SIGSEGV here; on the GPU the same overrun only faults when the neighbouring page happens to be unmapped (which is how
such bugs survive ordinary runs and then kill a profiled one).  HIPSIM_GUARD=1 keeps buffer starts 16-byte aligned (vector
paths), =2 ends buffers to 4 bytes at the guard page (scalar paths, exact to one int32), =3 guards the page BEFORE every buffer.
In every mode fresh memory and scratch handed out again are poisoned (0x7f bytes), so a kernel that consumes memory nobody wrote
uses an index ~2^31 elements away and faults too."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_guarded(mode: str, files, **extra):
    env = dict(os.environ, HIPSIM_GUARD=mode, **extra)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", *files], cwd=ROOT, env=env, capture_output=True, text=True)
    assert r.returncode == 0, f"HIPSIM_GUARD={mode}: rc {r.returncode}\n{r.stdout[-3000:]}\n{r.stderr[-2000:]}"


def test_guard_pages_catch_an_overrun(sim_lib):
    """The instrument itself: one element past a guarded input must fault."""
    probe = (
        "import sys, torch\n"
        f"sys.path[:0] = [{ROOT!r}, {os.path.join(ROOT, 'tests')!r}]\n"
        "from hostsim import build_sim\n"
        "from universal_recommender_amd import _lib\n"
        "from universal_recommender_amd.device import DeviceSession\n"
        "from conftest import guarded_tensor\n"
        "lib = _lib.load(build_sim.build())\n"
        "s = DeviceSession(torch.device('cpu'), lib)\n"
        "s.empty = lambda n, dtype: guarded_tensor(lib, n, dtype)\n"
        "n = 3072\n"
        "ci = guarded_tensor(lib, n, torch.int32); ci.copy_(torch.randint(0, 50, (n,), dtype=torch.int32))\n"
        "assert int(s.column_counts(ci, n, 50).sum()) == n\n"
        "print('exact ok', flush=True)\n"
        "s.column_counts(ci, n + int(sys.argv[1]), 50)\n"
        "print('overrun survived', flush=True)\n")
    for mode, over in (("2", "1"), ("1", "4")):
        r = subprocess.run([sys.executable, "-c", probe, over], env=dict(os.environ, HIPSIM_GUARD=mode), capture_output=True, text=True)
        assert "exact ok" in r.stdout and "overrun survived" not in r.stdout and r.returncode == -11, (mode, r.returncode, r.stdout, r.stderr[-500:])


def test_kernels_and_context_under_guard_pages_aligned(sim_lib):
    run_guarded("1", ["tests/test_sim_kernel_logic.py", "tests/test_sim_context.py"])


def test_kernels_under_guard_pages_exact(sim_lib):
    run_guarded("2", ["tests/test_sim_kernel_logic.py", "tests/test_sim_properties.py"])


def test_kernels_and_context_with_a_guard_page_before_every_buffer(sim_lib):
    """HIPSIM_GUARD=3: buffers START at a page boundary behind a PROT_NONE page -- buffer[-1], prefix[t - 1] at t = 0, a sentinel -1 used
    as an index (the one fault address recorded on hardware was the last page below a 2 MiB boundary)."""
    run_guarded("3", ["tests/test_sim_kernel_logic.py", "tests/test_sim_context.py"])


def test_kernels_with_the_waves_of_a_block_scheduled_in_reverse(sim_lib):
    """HIPSIM_ORDER=reverse: between two rendezvous the highest wave of a block runs first.  The default ascending order hides a
    missing __syncthreads() whenever the producing wave has the lower index -- which is also how hardware mostly happens to
    schedule, i.e. the kind of bug that survives ordinary runs and strikes once in a while."""
    run_guarded("1", ["tests/test_sim_kernel_logic.py", "tests/test_sim_properties.py"], HIPSIM_ORDER="reverse")


def test_kernels_with_lds_array_bounds_checked(sim_lib):
    """The kernel sources once more with -fsanitize=bounds (every index into a `__shared__` array checked; an out-of-range LDS
    access is dropped silently by ds_ instructions and is a memory aperture violation through flat ones), on top of guard pages."""
    run_guarded("1", ["tests/test_sim_kernel_logic.py"], HIPSIM_VARIANT="bounds")
