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
SOURCES = [os.path.join(CSRC, f) for f in ("cco_counts.hip", "cco_rowscan.hip", "cco_transpose.hip", "cco_expand.hip", "cco_rows.hip", "cco_misc.hip")] + [os.path.join(CSRC, "ingest_kernels.hip"), os.path.join(CSRC, "urcco_api.hip"), os.path.join(CSRC, "urcco_context.hip"), os.path.join(CSRC, "urcco_hash.hip"),
           os.path.join(HERE, "hipsim.cpp")]
DEPS = SOURCES + [os.path.join(CSRC, "cco_kernels.h"), os.path.join(CSRC, "cco_common.h"), os.path.join(CSRC, "cco_device.h"), os.path.join(CSRC, "urcco_internal.h"),
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


OUT_BOUNDS = os.path.join(HERE, "_build", "liburcco_hostsim_bounds.so")


def build_bounds(force: bool = False) -> str:
    """The same build with -fsanitize=bounds,object-size: every index into a `__shared__` array (a static array of known size
    here) is checked.  On the GPU an out-of-range LDS access through a ds_ instruction is silently dropped, through a flat
    instruction it is a memory aperture violation; in the plain simulator it would scribble over a neighbouring static."""
    if not force and os.path.exists(OUT_BOUNDS) and all(os.path.getmtime(OUT_BOUNDS) >= os.path.getmtime(d) for d in DEPS):
        return OUT_BOUNDS
    os.makedirs(os.path.dirname(OUT_BOUNDS), exist_ok=True)
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-fno-strict-aliasing", "-pthread",
           "-fsanitize=bounds,object-size", "-fno-sanitize-recover=all", "-I", os.path.join(HERE, "include"), "-Wno-unknown-pragmas"]
    for s in SOURCES:
        cmd += ["-x", "c++", s]
    cmd += ["-ldl", "-lubsan", "-o", OUT_BOUNDS]
    subprocess.check_call(cmd)
    return OUT_BOUNDS


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))


OUT_ASAN = os.path.join(HERE, "_build", "liburcco_hostsim_asan.so")
ORACLE_ASAN = os.path.join(HERE, "_build", "liburcco_oracle_asan.so")


def build_asan(force: bool = False):
    """The C-ABI translation units (urcco_api / urcco_context / urcco_hash: sessions, contexts, staging, validation, error
    paths) and the C oracle compiled with AddressSanitizer + UndefinedBehaviorSanitizer; the kernel sources and the fiber
    runtime stay uninstrumented (the simulator switches stacks by hand, which ASan's stack instrumentation cannot follow).
    Loaded by tests/test_sanitizers.py in a subprocess that preloads libasan."""
    os.makedirs(os.path.dirname(OUT_ASAN), exist_ok=True)
    if force or not os.path.exists(OUT_ASAN) or any(os.path.getmtime(OUT_ASAN) < os.path.getmtime(d) for d in DEPS):
        base = ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-fopenmp", "-ffp-contract=off", "-fno-strict-aliasing", "-pthread",
                "-I", os.path.join(HERE, "include"), "-Wno-unknown-pragmas"]
        san = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined"]
        objs = []
        for s in SOURCES:
            o = os.path.join(HERE, "_build", "asan_" + os.path.basename(s) + ".o")
            instrument = os.path.basename(s) in ("urcco_api.hip", "urcco_context.hip", "urcco_hash.hip")
            subprocess.check_call(base + (san if instrument else []) + ["-x", "c++", "-c", s, "-o", o])
            objs.append(o)
        subprocess.check_call(["g++", "-shared", "-fopenmp", "-pthread"] + san + objs + ["-ldl", "-o", OUT_ASAN])
    osrc = os.path.join(ROOT, "oracle", "cco_oracle.c")
    if force or not os.path.exists(ORACLE_ASAN) or os.path.getmtime(ORACLE_ASAN) < os.path.getmtime(osrc):
        subprocess.check_call(["gcc", "-O1", "-g", "-fPIC", "-std=c11", "-fopenmp", "-ffp-contract=off", "-fsanitize=address,undefined",
                               "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined", "-shared", osrc, "-lm", "-o", ORACLE_ASAN])
    return OUT_ASAN, ORACLE_ASAN
