// urcco_jni.cpp -- JNI shim between the Universal Recommender's Scala host and liburcco (include/urcco.h).
//
// Pure marshalling: it binds exactly the host-level entry points that replace the two Mahout calls of
// URAlgorithm.calcAll (reference src/main/scala/URAlgorithm.scala:323-329 `SimilarityAnalysis.cooccurrencesIDSs` and
// :343-346 `SimilarityAnalysis.crossOccurrenceDownsampled`); scala/HipSimilarityAnalysis.scala is the Scala side.
// The exported names are the JNI short names of the STATIC native methods of the JAVA class com.actionml.urcco.Native
// (java/com/actionml/urcco/Native.java) -- not of a Scala `object`, whose methods live on `Native$` (..._Native_00024_...).
//
// Build (on a box with a JDK):  make -C jni JAVA_HOME=/path/to/jdk      -> jni/liburcco_jni.so
// Checked here (no JDK in this image): `make -C jni check` compiles it against jni/stub/jni.h, and tests/test_jni_shim.py
// RUNS it against a fake JNIEnv (jni/test/fake_jvm.cpp) on top of the library.
#include <jni.h>

#include <cmath>
#include <cstdint>
#include <vector>

#include "urcco.h"

namespace {

void throw_runtime(JNIEnv* env, const char* msg) {
  jclass cls = env->FindClass("java/lang/RuntimeException");
  if (cls) env->ThrowNew(cls, msg);
}

}  // namespace

extern "C" {

// static native Object[] crossOccurrenceDownsampled(long[][] rowPtrs, int[][] colIdxs, long[] nCols, int[] maxElementsPerRow,
//     int[] maxInterestingElements, double[] minLlr /* NaN = None */, int seed, int device, int nGpus /* 0 = all */)
// returns Object[3 * n] = { long[] rowPtr, int[] colIdx, double[] llr } per dataset; throws RuntimeException on failure (the
// host may then fall back to Mahout).
JNIEXPORT jobjectArray JNICALL Java_com_actionml_urcco_Native_crossOccurrenceDownsampled(JNIEnv* env, jclass, jobjectArray rowPtrs, jobjectArray colIdxs,
                                                                                      jlongArray nCols, jintArray maxElementsPerRow,
                                                                                      jintArray maxInterestingElements, jdoubleArray minLlr, jint seed,
                                                                                      jint device, jint nGpus) {
  const jsize n = env->GetArrayLength(rowPtrs);
  if (n <= 0 || env->GetArrayLength(colIdxs) != n || env->GetArrayLength(nCols) != n || env->GetArrayLength(maxElementsPerRow) != n ||
      env->GetArrayLength(maxInterestingElements) != n || env->GetArrayLength(minLlr) != n) {
    throw_runtime(env, "urcco: argument arrays must have one entry per dataset");
    return nullptr;
  }
  std::vector<urcco_dataset> ds((size_t)n);
  std::vector<jlongArray> rp((size_t)n);
  std::vector<jintArray> ci((size_t)n);
  std::vector<jsize> ci_len((size_t)n);
  jlong* ncols = env->GetLongArrayElements(nCols, nullptr);
  jint* max_row = env->GetIntArrayElements(maxElementsPerRow, nullptr);
  jint* max_int = env->GetIntArrayElements(maxInterestingElements, nullptr);
  jdouble* mllr = env->GetDoubleArrayElements(minLlr, nullptr);
  for (jsize d = 0; d < n; ++d) {
    rp[(size_t)d] = (jlongArray)env->GetObjectArrayElement(rowPtrs, d);
    ci[(size_t)d] = (jintArray)env->GetObjectArrayElement(colIdxs, d);
    ci_len[(size_t)d] = env->GetArrayLength(ci[(size_t)d]);
    urcco_dataset& x = ds[(size_t)d];
    x.matrix.n_rows = env->GetArrayLength(rp[(size_t)d]) - 1;
    x.matrix.n_cols = ncols[d];
    x.max_elements_per_row = max_row[d];
    x.max_interesting_elements = max_int[d];
    x.has_min_llr = std::isnan(mllr[d]) ? 0 : 1;
    x.min_llr = x.has_min_llr ? mllr[d] : 0.0;
    x.reserved = 0;
  }
  env->ReleaseLongArrayElements(nCols, ncols, JNI_ABORT);
  env->ReleaseIntArrayElements(maxElementsPerRow, max_row, JNI_ABORT);
  env->ReleaseIntArrayElements(maxInterestingElements, max_int, JNI_ABORT);
  env->ReleaseDoubleArrayElements(minLlr, mllr, JNI_ABORT);
  // The matrices are pinned only while the library READS them: urcco_cross_occurrence_stage returns once every byte sits in the
  // library's pinned staging ring (the build is already running behind the copies), the critical sections are released, and
  // only then does the shim wait for the model (urcco_cross_occurrence_finish) -- the JVM's garbage collector is locked out for
  // the duration of a memcpy-speed pass over the inputs, not for the build and the download of the results.  No JNI call is
  // made between the first Get...Critical and the last Release...Critical.
  // (a NULL return leaves an OutOfMemoryError pending, and GetPrimitiveArrayCritical is itself a call -Xcheck:jni flags with an
  // exception pending: the loop stops at the first failure, the arrays behind it stay unpinned -- their pointers NULL, which the
  // release loop skips)
  for (jsize d = 0; d < n; ++d) {
    ds[(size_t)d].matrix.row_ptr = nullptr;
    ds[(size_t)d].matrix.col_idx = nullptr;
  }
  for (jsize d = 0; d < n; ++d) {
    ds[(size_t)d].matrix.row_ptr = (const int64_t*)env->GetPrimitiveArrayCritical(rp[(size_t)d], nullptr);
    if (!ds[(size_t)d].matrix.row_ptr) break;
    ds[(size_t)d].matrix.col_idx = (const int32_t*)env->GetPrimitiveArrayCritical(ci[(size_t)d], nullptr);
    if (!ds[(size_t)d].matrix.col_idx && ci_len[(size_t)d] > 0) break;
  }
  urcco_options opt = {};
  opt.device = device;
  opt.row_rate_mode = URCCO_ROW_RATE_MAHOUT_INT_DIV;
  opt.n_gpus = nGpus;
  // The Scala host re-inserts every row by column index (SequentialAccessSparseVector.setQuick) and the reference sorts by
  // score later (toStringMapRDD, package.scala:102): the order inside a returned row is never observed, so the library may
  // skip its ranking pass.
  opt.flags = URCCO_FLAG_UNORDERED_ROWS;
  std::vector<urcco_indicators> out((size_t)n);
  bool null_array = false;
  for (jsize d = 0; d < n; ++d) null_array = null_array || !ds[(size_t)d].matrix.row_ptr || (!ds[(size_t)d].matrix.col_idx && ci_len[(size_t)d] > 0);
  int st = null_array ? URCCO_OOM_HOST : urcco_cross_occurrence_stage(ds.data(), n, seed, &opt);
  for (jsize d = n; d-- > 0;) {  // release in reverse order of acquisition
    if (ds[(size_t)d].matrix.col_idx) env->ReleasePrimitiveArrayCritical(ci[(size_t)d], (void*)ds[(size_t)d].matrix.col_idx, JNI_ABORT);
    if (ds[(size_t)d].matrix.row_ptr) env->ReleasePrimitiveArrayCritical(rp[(size_t)d], (void*)ds[(size_t)d].matrix.row_ptr, JNI_ABORT);
  }
  if (null_array) {
    // GetPrimitiveArrayCritical returned NULL: the JVM has (normally) left an OutOfMemoryError pending, and no JNI call that may
    // throw -- FindClass, ThrowNew -- is allowed with an exception pending (-Xcheck:jni aborts on it): let that one propagate
    if (!env->ExceptionCheck()) throw_runtime(env, "urcco: the JVM could not pin an input array");
    return nullptr;
  }
  if (st == URCCO_OK) st = urcco_cross_occurrence_finish(out.data(), n, nullptr);
  if (st != URCCO_OK) {
    throw_runtime(env, urcco_last_error());
    return nullptr;
  }
  jobjectArray res = env->NewObjectArray(3 * n, env->FindClass("java/lang/Object"), nullptr);
  bool ok = res != nullptr;
  for (jsize d = 0; ok && d < n; ++d) {
    const urcco_indicators& o = out[(size_t)d];
    if (o.nnz > 0x7fffffffll || o.n_rows + 1 > 0x7fffffffll) {  // a Java array holds < 2^31 elements
      ok = false;
      break;
    }
    jlongArray r = env->NewLongArray((jsize)(o.n_rows + 1));
    jintArray c = env->NewIntArray((jsize)o.nnz);
    jdoubleArray v = env->NewDoubleArray((jsize)o.nnz);
    if (!r || !c || !v) {
      ok = false;
      break;
    }
    env->SetLongArrayRegion(r, 0, (jsize)(o.n_rows + 1), (const jlong*)o.row_ptr);
    if (o.nnz > 0) {
      env->SetIntArrayRegion(c, 0, (jsize)o.nnz, (const jint*)o.col_idx);
      env->SetDoubleArrayRegion(v, 0, (jsize)o.nnz, o.llr);
    }
    env->SetObjectArrayElement(res, 3 * d, r);
    env->SetObjectArrayElement(res, 3 * d + 1, c);
    env->SetObjectArrayElement(res, 3 * d + 2, v);
  }
  urcco_free_indicators(out.data(), n);
  if (!ok) {
    throw_runtime(env, "urcco: the indicator matrices do not fit Java arrays (or the JVM is out of memory)");
    return nullptr;
  }
  return res;
}

// static native int deviceCount()
JNIEXPORT jint JNICALL Java_com_actionml_urcco_Native_deviceCount(JNIEnv*, jclass) { return urcco_device_count(); }

// static native void shutdown(): frees the library's process-wide context (streams, scratch, pinned pools, RCCL communicators)
JNIEXPORT void JNICALL Java_com_actionml_urcco_Native_shutdown(JNIEnv*, jclass) { (void)urcco_shutdown(); }

}  // extern "C"
