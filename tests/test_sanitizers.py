"""the C-ABI host code (sessions, contexts, staging, validation, error paths) and the C oracle under
AddressSanitizer + UndefinedBehaviorSanitizer, driven through the simulator build in a subprocess that preloads libasan."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def test_c_abi_and_oracle_under_asan_ubsan():
    from hostsim import build_sim
    sim, orc = build_sim.build_asan()
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:verify_asan_link_order=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, os.path.join(HERE, "sanitizer_driver.py"), sim, orc], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "SANITIZED_RUN_OK" in out.stdout, out.stdout[-3000:] + out.stderr[-6000:]
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-6000:]
