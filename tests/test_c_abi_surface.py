"""The C-ABI library builds for gfx950 and exports every symbol include/urcco.h declares (no compute calls: there is no
GPU here); without a HIP device the product path fails loudly instead of falling back to a CPU path."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "urcco.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(urcco_[a-z0-9_]+)\s*\(", text)))


def test_header_binding_and_library_agree():
    import __graft_entry__ as g
    from universal_recommender_amd import _lib
    lib_path = g.build_hip()
    declared = _declared()
    assert len(declared) >= 20
    assert sorted(_lib.SYMBOLS) == declared, "ctypes binding and include/urcco.h disagree"
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib_path], text=True)
    exported = set(re.findall(r" T (urcco_[a-z0-9_]+)", out))
    assert set(declared) <= exported, sorted(set(declared) - exported)
    lib = _lib.load(lib_path)                      # dlopen + symbol resolution of every declared entry point
    assert lib.urcco_version() == 305
    # the device code object really targets gfx950
    note = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", lib_path], capture_output=True, text=True).stdout
    blob = open(lib_path, "rb").read()
    assert b"gfx950" in blob


def test_no_device_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import __graft_entry__ as g
    from universal_recommender_amd import _lib
    from universal_recommender_amd.device import DeviceSession
    lib = _lib.load(g.build_hip())
    assert lib.urcco_device_count() == 0
    with pytest.raises(_lib.UrccoError) as ei:
        DeviceSession(torch.device("cpu"), lib)
    assert ei.value.status == _lib.NO_DEVICE


def test_missing_library_is_an_error():
    from universal_recommender_amd import _lib
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load("/nonexistent/liburcco.so")


def test_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "universal-recommender_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower() or f == "__init__.py", f"{f} mentions the oracle"
