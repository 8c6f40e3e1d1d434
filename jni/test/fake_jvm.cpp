// TEST-ONLY fake JVM: implements the JNIEnv of jni/stub/jni.h on plain heap objects and drives the shim's native method
// the way the Scala host does (Native.crossOccurrenceDownsampled).  Like a JVM it LINKS the native methods by name: the
// caller declares the holder class and the method (fake_jvm_bind), the name is mangled as the JNI specification prescribes
// ('.' -> '_', '_' -> "_1", ';' -> "_2", '[' -> "_3", anything else outside [A-Za-z0-9] -> "_0xxxx": '$' of a Scala module
// class becomes "_00024") and resolved with dlsym in this very library -- an unresolved name is the UnsatisfiedLinkError a
// real JVM throws at the first call (round 5's `object Native` would have been caught here).  tests/test_jni_shim.py loads this through ctypes,
// hands it numpy CSR matrices and compares what comes back with the oracle -- so the marshalling of jni/urcco_jni.cpp is
// executed, not only type-checked, although this image has no JDK.  It also enforces the JNI critical-section rule: no
// JNI call may be made while a primitive array is held critically.
#include <jni.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

using CrossFn = jobjectArray (*)(JNIEnv*, jclass, jobjectArray, jobjectArray, jlongArray, jintArray, jintArray, jdoubleArray, jint, jint, jint);
using CountFn = jint (*)(JNIEnv*, jclass);
using ShutFn = void (*)(JNIEnv*, jclass);
CrossFn g_cross = nullptr;  // bound by fake_jvm_bind, as a JVM binds a native method at its first call
CountFn g_count = nullptr;
ShutFn g_shut = nullptr;

// JNI specification, "Resolving Native Method Names": Java_ + mangled fully-qualified class name + _ + mangled method name
std::string jni_mangle(const std::string& s) {
  std::string out;
  for (unsigned char c : s) {
    if (c == '.' || c == '/') out += '_';
    else if (c == '_') out += "_1";
    else if (c == ';') out += "_2";
    else if (c == '[') out += "_3";
    else if ((c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z')) out += (char)c;
    else {
      char buf[8];
      snprintf(buf, sizeof buf, "_0%04x", (unsigned)c);
      out += buf;
    }
  }
  return out;
}

void* resolve_native(const char* cls, const char* method, std::string* symbol) {
  *symbol = "Java_" + jni_mangle(cls) + "_" + jni_mangle(method);
  Dl_info info;
  if (!dladdr((void*)&jni_mangle, &info) || !info.dli_fname) return nullptr;
  void* self = dlopen(info.dli_fname, RTLD_NOW | RTLD_NOLOAD);  // the library System.loadLibrary would have loaded: the shim is linked into it
  if (!self) return nullptr;
  void* f = dlsym(self, symbol->c_str());
  dlclose(self);
  return f;
}

struct IntArr : _jintArray { std::vector<jint> v; };
struct LongArr : _jlongArray { std::vector<jlong> v; };
struct DoubleArr : _jdoubleArray { std::vector<jdouble> v; };
struct ObjArr : _jobjectArray { std::vector<jobject> v; };
struct Cls : _jclass { std::string name; };

struct FakeEnv : JNIEnv {
  std::vector<_jobject*> heap;
  int critical = 0;          // primitive arrays currently held critically
  int violations = 0;        // JNI calls made inside a critical section
  std::string pending;       // message of the pending exception ("" = none)
  int fail_pin_after = -1;   // >= 0: the (fail_pin_after + 1)-th GetPrimitiveArrayCritical fails, leaving an OutOfMemoryError pending
  int pins = 0;
  int pending_violations = 0;  // JNI calls that may throw, made with an exception already pending (-Xcheck:jni aborts on them)
  template <typename T> T* keep(T* o) { heap.push_back(o); return o; }
  ~FakeEnv() override { for (_jobject* o : heap) delete o; }
  void call() { if (critical > 0) ++violations; }

  jclass FindClass(const char* name) override { call(); if (!pending.empty()) ++pending_violations; Cls* c = keep(new Cls()); c->name = name; return c; }
  jint ThrowNew(jclass, const char* msg) override { call(); if (!pending.empty()) ++pending_violations; pending = msg ? msg : "(null)"; return 0; }
  jboolean ExceptionCheck() override { return pending.empty() ? 0 : 1; }  // allowed with an exception pending, and inside a critical section
  jsize GetArrayLength(jarray a) override {
    call();
    if (auto* x = dynamic_cast<IntArr*>(a)) return (jsize)x->v.size();
    if (auto* x = dynamic_cast<LongArr*>(a)) return (jsize)x->v.size();
    if (auto* x = dynamic_cast<DoubleArr*>(a)) return (jsize)x->v.size();
    if (auto* x = dynamic_cast<ObjArr*>(a)) return (jsize)x->v.size();
    return -1;
  }
  jobject GetObjectArrayElement(jobjectArray a, jsize i) override { call(); return static_cast<ObjArr*>(a)->v.at((size_t)i); }
  void SetObjectArrayElement(jobjectArray a, jsize i, jobject v) override { call(); static_cast<ObjArr*>(a)->v.at((size_t)i) = v; }
  jobjectArray NewObjectArray(jsize n, jclass, jobject init) override { call(); ObjArr* o = keep(new ObjArr()); o->v.assign((size_t)n, init); return o; }
  jintArray NewIntArray(jsize n) override { call(); IntArr* o = keep(new IntArr()); o->v.assign((size_t)n, 0); return o; }
  jlongArray NewLongArray(jsize n) override { call(); LongArr* o = keep(new LongArr()); o->v.assign((size_t)n, 0); return o; }
  jdoubleArray NewDoubleArray(jsize n) override { call(); DoubleArr* o = keep(new DoubleArr()); o->v.assign((size_t)n, 0.0); return o; }
  jint* GetIntArrayElements(jintArray a, jboolean* c) override { call(); if (c) *c = 0; return static_cast<IntArr*>(a)->v.data(); }
  jlong* GetLongArrayElements(jlongArray a, jboolean* c) override { call(); if (c) *c = 0; return static_cast<LongArr*>(a)->v.data(); }
  jdouble* GetDoubleArrayElements(jdoubleArray a, jboolean* c) override { call(); if (c) *c = 0; return static_cast<DoubleArr*>(a)->v.data(); }
  void ReleaseIntArrayElements(jintArray, jint*, jint) override { call(); }
  void ReleaseLongArrayElements(jlongArray, jlong*, jint) override { call(); }
  void ReleaseDoubleArrayElements(jdoubleArray, jdouble*, jint) override { call(); }
  void SetIntArrayRegion(jintArray a, jsize s, jsize n, const jint* b) override { call(); memcpy(static_cast<IntArr*>(a)->v.data() + s, b, sizeof(jint) * (size_t)n); }
  void SetLongArrayRegion(jlongArray a, jsize s, jsize n, const jlong* b) override { call(); memcpy(static_cast<LongArr*>(a)->v.data() + s, b, sizeof(jlong) * (size_t)n); }
  void SetDoubleArrayRegion(jdoubleArray a, jsize s, jsize n, const jdouble* b) override {
    call();
    memcpy(static_cast<DoubleArr*>(a)->v.data() + s, b, sizeof(jdouble) * (size_t)n);
  }
  void* GetPrimitiveArrayCritical(jarray a, jboolean* c) override {  // allowed inside a critical section
    if (!pending.empty()) ++pending_violations;  // ... but not with an exception pending (the shim kept pinning after a failed pin)
    if (fail_pin_after >= 0 && pins++ == fail_pin_after) {
      pending = "java.lang.OutOfMemoryError";
      return nullptr;
    }
    ++critical;
    if (c) *c = 0;
    if (auto* x = dynamic_cast<IntArr*>(a)) return x->v.data();
    if (auto* x = dynamic_cast<LongArr*>(a)) return x->v.data();
    if (auto* x = dynamic_cast<DoubleArr*>(a)) return x->v.data();
    return nullptr;
  }
  // A released array may be moved by the garbage collector at once: the fake JVM overwrites it, so a library that still read
  // the caller's matrices after the shim released them (the release now comes BEFORE the build is awaited) would compute garbage
  void ReleasePrimitiveArrayCritical(jarray a, void*, jint) override {
    --critical;
    if (auto* x = dynamic_cast<IntArr*>(a)) std::fill(x->v.begin(), x->v.end(), (jint)0x5a5a5a5a);
    if (auto* x = dynamic_cast<LongArr*>(a)) std::fill(x->v.begin(), x->v.end(), (jlong)0x5a5a5a5a5a5a5a5all);
  }
};

}  // namespace

extern "C" {

// Runs Native.crossOccurrenceDownsampled on n datasets.  Outputs are malloc'ed (release with fake_jvm_free):
// out_row_ptr[d] (n_items_a + 1), out_col_idx[d], out_llr[d] (out_nnz[d] entries).  Returns 0, or 1 with the pending
// RuntimeException's message in err, or 2 on a JNI rule violation.
int fake_jvm_cross_occurrence(int n, const int64_t* n_rows, const int64_t* n_cols, const int64_t* const* row_ptr, const int32_t* const* col_idx,
                              const int32_t* max_rows, const int32_t* max_int, const double* min_llr, int seed, int device, int n_gpus,
                              int64_t** out_row_ptr, int32_t** out_col_idx, double** out_llr, int64_t* out_nnz, int64_t* out_rows, char* err,
                              int err_cap) {
  FakeEnv env;
  if (const char* e = getenv("FAKE_JVM_FAIL_PIN_AFTER")) env.fail_pin_after = atoi(e);
  ObjArr* rps = env.keep(new ObjArr());
  ObjArr* cis = env.keep(new ObjArr());
  LongArr* nc = env.keep(new LongArr());
  IntArr* mr = env.keep(new IntArr());
  IntArr* mi = env.keep(new IntArr());
  DoubleArr* ml = env.keep(new DoubleArr());
  for (int d = 0; d < n; ++d) {
    LongArr* rp = env.keep(new LongArr());
    rp->v.assign(row_ptr[d], row_ptr[d] + n_rows[d] + 1);
    IntArr* ci = env.keep(new IntArr());
    ci->v.assign(col_idx[d], col_idx[d] + row_ptr[d][n_rows[d]]);
    rps->v.push_back(rp);
    cis->v.push_back(ci);
    nc->v.push_back(n_cols[d]);
    mr->v.push_back(max_rows[d]);
    mi->v.push_back(max_int[d]);
    ml->v.push_back(min_llr[d]);
  }
  if (!g_cross) {
    snprintf(err, (size_t)err_cap, "java.lang.UnsatisfiedLinkError: crossOccurrenceDownsampled is not bound (call fake_jvm_bind first)");
    return 3;
  }
  jobjectArray res = g_cross(&env, nullptr, rps, cis, nc, mr, mi, ml, seed, device, n_gpus);
  if (env.violations > 0 || env.critical != 0 || env.pending_violations > 0) {
    snprintf(err, (size_t)err_cap, "JNI rule violated: %d call(s) inside a critical section, %d array(s) still held, %d throwing call(s) with an exception pending",
             env.violations, env.critical, env.pending_violations);
    return 2;
  }
  if (!env.pending.empty() || !res) {
    snprintf(err, (size_t)err_cap, "%s", env.pending.empty() ? "null result without exception" : env.pending.c_str());
    return 1;
  }
  ObjArr* r = static_cast<ObjArr*>(res);
  for (int d = 0; d < n; ++d) {
    LongArr* rp = static_cast<LongArr*>(r->v.at((size_t)3 * d));
    IntArr* ci = static_cast<IntArr*>(r->v.at((size_t)3 * d + 1));
    DoubleArr* ll = static_cast<DoubleArr*>(r->v.at((size_t)3 * d + 2));
    out_rows[d] = (int64_t)rp->v.size() - 1;
    out_nnz[d] = (int64_t)ci->v.size();
    out_row_ptr[d] = (int64_t*)malloc(sizeof(int64_t) * rp->v.size());
    out_col_idx[d] = (int32_t*)malloc(sizeof(int32_t) * (ci->v.size() + 1));
    out_llr[d] = (double*)malloc(sizeof(double) * (ll->v.size() + 1));
    memcpy(out_row_ptr[d], rp->v.data(), sizeof(int64_t) * rp->v.size());
    memcpy(out_col_idx[d], ci->v.data(), sizeof(int32_t) * ci->v.size());
    memcpy(out_llr[d], ll->v.data(), sizeof(double) * ll->v.size());
  }
  return 0;
}

void fake_jvm_free(void* p) { free(p); }
int fake_jvm_device_count(void) { return g_count ? g_count(nullptr, nullptr) : -1; }
void fake_jvm_shutdown(void) { if (g_shut) g_shut(nullptr, nullptr); }

// 1 when `cls`.`method` (cls = the JVM's binary class name, e.g. "com.actionml.urcco.Native" or "...Native$") links against
// the shim by the JNI short name; the mangled symbol that was looked up is written to `symbol`.
int fake_jvm_resolves(const char* cls, const char* method, char* symbol, int symbol_cap) {
  std::string sym;
  void* f = resolve_native(cls, method, &sym);
  if (symbol && symbol_cap > 0) snprintf(symbol, (size_t)symbol_cap, "%s", sym.c_str());
  return f ? 1 : 0;
}

// Links the three native methods of the holder class `cls`; 0, or 3 with "java.lang.UnsatisfiedLinkError: <symbol>" in err.
int fake_jvm_bind(const char* cls, char* err, int err_cap) {
  std::string sym;
  g_cross = (CrossFn)resolve_native(cls, "crossOccurrenceDownsampled", &sym);
  if (g_cross) g_count = (CountFn)resolve_native(cls, "deviceCount", &sym);
  if (g_cross && g_count) g_shut = (ShutFn)resolve_native(cls, "shutdown", &sym);
  if (g_cross && g_count && g_shut) return 0;
  g_cross = nullptr, g_count = nullptr, g_shut = nullptr;
  snprintf(err, (size_t)err_cap, "java.lang.UnsatisfiedLinkError: %s", sym.c_str());
  return 3;
}

}  // extern "C"
