"""The JNI shim (jni/urcco_jni.cpp) without a JVM: `make -C jni check` type-checks it against the stub <jni.h>, and a fake
JNIEnv (jni/test/fake_jvm.cpp) RUNS Native.crossOccurrenceDownsampled on top of the library -- the simulator build here,
the product library on the GPU box -- so that the marshalling (critical sections released after the staging half of the
call -- the fake JVM overwrites a released array, so a library that read it later would fail --, NaN = None, Object[3n]
result, RuntimeException on failure) is executed and its output compared with the oracle."""
import ctypes as C
import glob
import os
import re
import subprocess

import numpy as np
import pytest

from helpers import check_indicators, rand_csr, sort_rows
from oracle import c_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JNI = os.path.join(ROOT, "jni")


def test_shim_type_checks_against_the_stub_header():
    subprocess.check_call(["make", "-s", "-C", JNI, "check"])


def declared_native_methods():
    """(JVM binary class name, method) of every native method the host sources declare -- read from the sources a maintainer
    copies into the reference tree, not from the shim.  A Java `class X` holds them on X; a Scala `object X` holds them on
    the module class X$ (round 5's defect: the shim exported ..._Native_... while the JVM looked for ..._Native_00024_...)."""
    out = []
    for path in glob.glob(os.path.join(ROOT, "java", "**", "*.java"), recursive=True):
        src = re.sub(r"/\*.*?\*/|//[^\n]*", "", open(path).read(), flags=re.S)
        pkg = re.search(r"\bpackage\s+([\w.]+)\s*;", src).group(1)
        cls = re.search(r"\bclass\s+(\w+)", src).group(1)
        for m in re.finditer(r"\b(static\s+)?native\b[^;(]*?\b(\w+)\s*\(", src):
            assert m.group(1), f"{m.group(2)}: the shim's functions take a jclass, i.e. bind STATIC native methods"
            out.append((f"{pkg}.{cls}", m.group(2)))
    for path in glob.glob(os.path.join(ROOT, "scala", "*.scala")):
        src = re.sub(r"/\*.*?\*/|//[^\n]*", "", open(path).read(), flags=re.S)
        pkg = re.search(r"\bpackage\s+([\w.]+)", src).group(1)
        for m in re.finditer(r"@native\s+def\s+(\w+)", src):
            holders = list(re.finditer(r"\b(object|class)\s+(\w+)", src[:m.start()]))
            kind, name = holders[-1].group(1), holders[-1].group(2)
            out.append((f"{pkg}.{name}" + ("$" if kind == "object" else ""), m.group(1)))
    return out


def _fake_jvm(lib_path: str, tag: str):
    subprocess.check_call(["make", "-s", "-C", JNI, "fake", f"URCCO_SO={lib_path}", f"TAG={tag}"])
    jvm = C.CDLL(os.path.join(JNI, "_build", f"libfake_jvm_{tag}.so"))
    # link like a JVM: by the mangled name of the class that DECLARES the native methods
    holders = sorted({c for c, _ in declared_native_methods()})
    assert len(holders) == 1, holders
    err = C.create_string_buffer(600)
    assert jvm.fake_jvm_bind(holders[0].encode(), err, 600) == 0, err.value.decode()
    return jvm


def test_native_methods_resolve_by_jni_name(sim_lib):
    """Every native method the host sources declare links against the shim under the name a JVM computes for it; the same
    methods on a Scala module class (`object Native` -> Native$ -> ..._Native_00024_...) do not: the negative control."""
    from hostsim import build_sim
    jvm = _fake_jvm(build_sim.OUT, "sim")
    decl = declared_native_methods()
    assert sorted(m for _, m in decl) == ["crossOccurrenceDownsampled", "deviceCount", "shutdown"], decl
    sym = C.create_string_buffer(300)
    for cls, method in decl:
        assert not cls.endswith("$"), f"{cls}: native methods of a Scala object bind to the module class; keep them on the Java holder"
        assert jvm.fake_jvm_resolves(cls.encode(), method.encode(), sym, 300) == 1, f"UnsatisfiedLinkError: {sym.value.decode()}"
        assert sym.value.decode() == "Java_" + cls.replace(".", "_") + "_" + method
        assert jvm.fake_jvm_resolves((cls + "$").encode(), method.encode(), sym, 300) == 0
        assert "_00024_" in sym.value.decode()
    err = C.create_string_buffer(600)
    assert jvm.fake_jvm_bind(b"com.actionml.urcco.Native$", err, 600) == 3 and b"UnsatisfiedLinkError" in err.value
    assert jvm.fake_jvm_device_count() == -1                       # nothing is bound after a failed link
    assert jvm.fake_jvm_bind(decl[0][0].encode(), err, 600) == 0


def test_scala_host_keeps_mahouts_signatures():
    """the two entry points keep Mahout 0.13.0's parameter lists (names matter: URAlgorithm.scala:323-329 calls
    with named arguments), incl. the ignored `parOpts`."""
    src = open(os.path.join(ROOT, "scala", "HipSimilarityAnalysis.scala")).read()
    sig = re.search(r"def cooccurrencesIDSs\((.*?)\)\s*:\s*List\[IndexedDataset\]", src, flags=re.S).group(1)
    names = re.findall(r"(\w+)\s*:", sig)
    assert names == ["indexedDatasets", "randomSeed", "maxInterestingItemsPerThing", "maxNumInteractions", "parOpts"], names
    assert "parOpts: ParOpts = defaultParOpts" in sig
    sig = re.search(r"def crossOccurrenceDownsampled\((.*?)\)\s*:\s*List\[IndexedDataset\]", src, flags=re.S).group(1)
    assert re.findall(r"(\w+)\s*:", sig) == ["datasets", "randomSeed"]
    assert "rp(nrow).toInt" not in src.replace("new Array[Int](rp(nrow).toInt)", "") and "Int.MaxValue" in src  # the 2^31 guard precedes the cast


def run_shim(jvm, mats, params, seed, n_gpus=1):
    n = len(mats)
    p64, p32 = C.POINTER(C.c_int64), C.POINTER(C.c_int32)
    rows = (C.c_int64 * n)(*[m.n_rows for m in mats])
    cols = (C.c_int64 * n)(*[m.n_cols for m in mats])
    rps = (p64 * n)(*[m.row_ptr.ctypes.data_as(p64) for m in mats])
    cis = (p32 * n)(*[(m.col_idx if m.nnz else np.zeros(1, np.int32)).ctypes.data_as(p32) for m in mats])
    mr = (C.c_int32 * n)(*[p.max_elements_per_row for p in params])
    mi = (C.c_int32 * n)(*[p.max_interesting_elements for p in params])
    ml = (C.c_double * n)(*[float("nan") if p.min_llr is None else p.min_llr for p in params])
    o_rp, o_ci, o_ll = (p64 * n)(), (p32 * n)(), (C.POINTER(C.c_double) * n)()
    o_nnz, o_rows = (C.c_int64 * n)(), (C.c_int64 * n)()
    err = C.create_string_buffer(600)
    st = jvm.fake_jvm_cross_occurrence(n, rows, cols, rps, cis, mr, mi, ml, seed, 0, n_gpus, o_rp, o_ci, o_ll, o_nnz, o_rows, err, 600)
    if st != 0:
        return st, err.value.decode(), None
    out = []
    for d in range(n):
        nnz = int(o_nnz[d])
        out.append((np.ctypeslib.as_array(o_rp[d], shape=(int(o_rows[d]) + 1,)).copy(), np.ctypeslib.as_array(o_ci[d], shape=(max(nnz, 1),))[:nnz].copy(),
                    np.ctypeslib.as_array(o_ll[d], shape=(max(nnz, 1),))[:nnz].copy()))
        for ptr in (o_rp[d], o_ci[d], o_ll[d]):
            jvm.fake_jvm_free(C.cast(ptr, C.c_void_p))
    return 0, "", out


def shim_case(jvm):
    jvm.fake_jvm_free.argtypes = [C.c_void_p]
    rng = np.random.default_rng(3)
    mats = [rand_csr(rng, 2000, 500, 9, zipf_s=1.1), rand_csr(rng, 2000, 800, 14), rand_csr(rng, 2000, 12, 2, empty_frac=0.3)]
    params = [O.DatasetParams(30, 10, None), O.DatasetParams(40, 12, 0.4), O.DatasetParams(500, 50, None)]
    ref = O.cross_occurrence_downsampled(mats, params, -559038737)
    for _ in range(2):
        st, msg, out = run_shim(jvm, mats, params, -559038737)
        assert st == 0, msg
        for got, r in zip(out, ref):
            check_indicators(sort_rows(got), r)      # the shim asks for URCCO_FLAG_UNORDERED_ROWS: rows are top-k SETS
    # failure -> RuntimeException carrying urcco_last_error, arrays released, no JNI call inside the critical section
    bad = O.Csr(2000, 4, mats[0].row_ptr, mats[0].col_idx)             # columns out of range
    st, msg, _ = run_shim(jvm, [bad], [params[0]], 1)
    assert st == 1 and "invalid entries" in msg, (st, msg)
    st, msg, _ = run_shim(jvm, mats[:1], [O.DatasetParams(0, 10, None)], 1)
    assert st == 1 and "positive" in msg, (st, msg)
    # the JVM cannot pin an array (GetPrimitiveArrayCritical -> NULL, OutOfMemoryError pending): everything pinned so far is
    # released and the pending error propagates -- no FindClass / ThrowNew on top of it (-Xcheck:jni aborts on that)
    for k in (0, 1, 2):
        os.environ["FAKE_JVM_FAIL_PIN_AFTER"] = str(k)
        try:
            st, msg, _ = run_shim(jvm, mats[:2], params[:2], 1)
        finally:
            del os.environ["FAKE_JVM_FAIL_PIN_AFTER"]
        assert st == 1 and msg == "java.lang.OutOfMemoryError", (k, st, msg)
    assert jvm.fake_jvm_device_count() >= 1
    jvm.fake_jvm_shutdown()
    st, msg, out = run_shim(jvm, mats[:2], params[:2], 7)              # the context is re-created after shutdown
    assert st == 0, msg
    jvm.fake_jvm_shutdown()


def test_shim_runs_against_a_fake_jvm(sim_lib):
    from hostsim import build_sim
    shim_case(_fake_jvm(build_sim.OUT, "sim"))


@pytest.mark.gpu
def test_shim_runs_against_a_fake_jvm_on_gpu(gpu_session):
    from universal_recommender_amd import _lib
    shim_case(_fake_jvm(_lib.DEFAULT_PATH, "gpu"))
