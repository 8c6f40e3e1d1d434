"""TEST-ONLY: compile the unchanged kernel sources against the host simulator (tests/hostsim/include).

The result (tests/hostsim/_build/liburcco_hostsim.so) exports the same C ABI as the product library but is
never loaded by the package: only CPU tests that exercise kernel/host LOGIC load it explicitly.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "universal-recommender_amd", "csrc")
OUT = os.path.join(HERE, "_build", "liburcco_hostsim.so")
SOURCES = [os.path.join(CSRC, "cco_kernels.hip"), os.path.join(CSRC, "ingest_kernels.hip"), os.path.join(CSRC, "urcco_api.hip"), os.path.join(CSRC, "urcco_context.hip"), os.path.join(CSRC, "urcco_hash.hip"),
           os.path.join(HERE, "hipsim.cpp")]
DEPS = SOURCES + [os.path.join(CSRC, "cco_kernels.h"), os.path.join(CSRC, "cco_device.h"), os.path.join(CSRC, "urcco_internal.h"),
                  os.path.join(ROOT, "include", "urcco.h"), os.path.join(HERE, "include", "hip", "hip_runtime.h")]


def build(force: bool = False) -> str:
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = ["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-fno-strict-aliasing", "-pthread",
           "-I", os.path.join(HERE, "include"), "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas", "-Wno-unused-variable"]
    for s in SOURCES:
        cmd += ["-x", "c++", s]
    cmd += ["-ldl", "-o", OUT]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
