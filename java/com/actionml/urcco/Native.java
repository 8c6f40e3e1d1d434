/*
 * Native -- the JVM holder of liburcco_jni.so's native methods (jni/urcco_jni.cpp).
 *
 * A plain Java class on purpose: `static native` methods of class com.actionml.urcco.Native resolve to the exported
 * symbols Java_com_actionml_urcco_Native_<method>.  (Round 5 declared them as `@native def` inside a Scala `object Native`:
 * those live on the module class `Native$`, for which the JVM looks up Java_com_actionml_urcco_Native_00024_<method> --
 * an UnsatisfiedLinkError at the first `pio train`.)  sbt compiles src/main/java beside src/main/scala; copy this file to
 * src/main/java/com/actionml/urcco/Native.java of the reference tree (INTEGRATION.md section 4).
 * tests/test_jni_shim.py reads THIS file, mangles class + method names the way the JNI specification does and resolves
 * the result in the shim with dlsym before it calls anything.
 */
package com.actionml.urcco;

public final class Native {
  static {
    System.loadLibrary("urcco_jni");
  }

  private Native() {}

  /**
   * @return Object[3 * n]: { long[] rowPtr, int[] colIdx, double[] llr } per dataset; throws RuntimeException when the
   *         library reports an error (the caller may fall back to Mahout).  minLlr: NaN = None.  nGpus: 0 = every visible GPU.
   */
  public static native Object[] crossOccurrenceDownsampled(long[][] rowPtrs, int[][] colIdxs, long[] nCols, int[] maxElementsPerRow,
      int[] maxInterestingElements, double[] minLlr, int seed, int device, int nGpus);

  public static native int deviceCount();

  public static native void shutdown();
}
