/* jni/stub/jni.h -- TEST-ONLY stand-in for the JDK's <jni.h>.
 *
 * This image has no JDK, so the shim (jni/urcco_jni.cpp) cannot be compiled against the real header here.  This file
 * declares the subset of the JNI C++ interface the shim uses, with the type lattice the JNI specification prescribes
 * (jlongArray -> jarray -> jobject, ...), so that `make -C jni check` type-checks the shim and jni/test/fake_jvm.cpp can
 * run it.  Unlike the real header, JNIEnv's members are virtual here: the fake JVM of the tests implements them.
 * Never shipped, never on the include path of a real build (the Makefile uses $(JAVA_HOME)/include when it is set). */
#ifndef URCCO_STUB_JNI_H
#define URCCO_STUB_JNI_H
#include <stdint.h>

typedef int32_t jint;
typedef int64_t jlong;
typedef double jdouble;
typedef uint8_t jboolean;
typedef jint jsize;

class _jobject { public: virtual ~_jobject() {} };
class _jclass : public _jobject {};
class _jthrowable : public _jobject {};
class _jarray : public _jobject {};
class _jobjectArray : public _jarray {};
class _jintArray : public _jarray {};
class _jlongArray : public _jarray {};
class _jdoubleArray : public _jarray {};
typedef _jobject* jobject;
typedef _jclass* jclass;
typedef _jthrowable* jthrowable;
typedef _jarray* jarray;
typedef _jobjectArray* jobjectArray;
typedef _jintArray* jintArray;
typedef _jlongArray* jlongArray;
typedef _jdoubleArray* jdoubleArray;

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_OK 0
#define JNI_COMMIT 1
#define JNI_ABORT 2

struct JNIEnv {
  virtual ~JNIEnv() {}
  virtual jclass FindClass(const char* name) = 0;
  virtual jint ThrowNew(jclass cls, const char* msg) = 0;
  virtual jboolean ExceptionCheck() = 0;
  virtual jsize GetArrayLength(jarray a) = 0;
  virtual jobject GetObjectArrayElement(jobjectArray a, jsize i) = 0;
  virtual void SetObjectArrayElement(jobjectArray a, jsize i, jobject v) = 0;
  virtual jobjectArray NewObjectArray(jsize n, jclass cls, jobject init) = 0;
  virtual jintArray NewIntArray(jsize n) = 0;
  virtual jlongArray NewLongArray(jsize n) = 0;
  virtual jdoubleArray NewDoubleArray(jsize n) = 0;
  virtual jint* GetIntArrayElements(jintArray a, jboolean* is_copy) = 0;
  virtual jlong* GetLongArrayElements(jlongArray a, jboolean* is_copy) = 0;
  virtual jdouble* GetDoubleArrayElements(jdoubleArray a, jboolean* is_copy) = 0;
  virtual void ReleaseIntArrayElements(jintArray a, jint* p, jint mode) = 0;
  virtual void ReleaseLongArrayElements(jlongArray a, jlong* p, jint mode) = 0;
  virtual void ReleaseDoubleArrayElements(jdoubleArray a, jdouble* p, jint mode) = 0;
  virtual void SetIntArrayRegion(jintArray a, jsize start, jsize len, const jint* buf) = 0;
  virtual void SetLongArrayRegion(jlongArray a, jsize start, jsize len, const jlong* buf) = 0;
  virtual void SetDoubleArrayRegion(jdoubleArray a, jsize start, jsize len, const jdouble* buf) = 0;
  virtual void* GetPrimitiveArrayCritical(jarray a, jboolean* is_copy) = 0;
  virtual void ReleasePrimitiveArrayCritical(jarray a, void* p, jint mode) = 0;
};
#endif
