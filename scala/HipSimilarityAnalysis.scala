/*
 * HipSimilarityAnalysis -- the Scala host side of liburcco: Mahout-identical signatures for the two calls
 * URAlgorithm.calcAll makes (reference src/main/scala/URAlgorithm.scala:323-329 and :343-346), implemented on the
 * MI355X through the JNI shim jni/urcco_jni.cpp -> include/urcco.h.
 *
 * Drop-in: in URAlgorithm.scala replace `SimilarityAnalysis.` by `HipSimilarityAnalysis.` at the two call sites (see
 * INTEGRATION.md section 4); Preparator.prepare before and `.map(_.asInstanceOf[IndexedDatasetSpark])` / URModel.save after
 * stay untouched.  Cannot be compiled in the build image of this repository (no JDK / Scala / Mahout 0.13.0 jars); the
 * native half is exercised by tests/test_jni_shim.py against a fake JNIEnv.
 *
 * The native methods live on the JAVA class com.actionml.urcco.Native (java/com/actionml/urcco/Native.java): `static native`
 * methods of a Java class bind to Java_com_actionml_urcco_Native_<method>, which is what the shim exports.  Do NOT move them
 * into a Scala `object`: its methods belong to the module class `Native$` and the JVM would look for
 * Java_com_actionml_urcco_Native_00024_<method> (tests/test_jni_shim.py::test_native_methods_resolve_by_jni_name guards this).
 */
package com.actionml.urcco

import org.apache.mahout.math.{SequentialAccessSparseVector, Vector}
import org.apache.mahout.math.cf.{DownsamplableCrossOccurrenceDataset, ParOpts}
import org.apache.mahout.math.indexeddataset.IndexedDataset
import org.apache.mahout.sparkbindings.{DrmRdd, SparkDistributedContext, drmWrap}
import org.apache.mahout.sparkbindings.indexeddataset.IndexedDatasetSpark

object HipSimilarityAnalysis {

  /** engine.json `numGPUs` (additive optional key); 0 = every GPU of the node */
  @volatile var numGPUs: Int = 0

  /** Mahout's SimilarityAnalysis.defaultParOpts; accepted for signature compatibility, meaningless off Spark */
  val defaultParOpts: ParOpts = ParOpts()

  /** One Spark partition of a DRM as three primitive arrays: row indices, row lengths, concatenated sorted column indices */
  private final class CsrChunk(val rows: Array[Int], val lens: Array[Int], val cols: Array[Int]) extends Serializable

  /** The driver materialises a DRM as CSR: (rowIdx, sorted non-zero column indices).  Values are all 1.0
    * (Preparator.scala:146,205), so only the structure travels; users without a row in this event type get an empty row.
    * Executors flatten their partition into PRIMITIVE arrays (no boxed Array[Int] per row reaches the driver: at BASELINE
    * config 4 a matrix has 10 M rows / 400 M entries), the driver places the chunks with System.arraycopy. */
  private def toCsr(ids: IndexedDataset): (Array[Long], Array[Int]) = {
    val nrowL: Long = ids.matrix.nrow
    require(nrowL < Int.MaxValue, s"urcco: ${nrowL} rows do not fit a Java array")
    val nrow = nrowL.toInt
    val chunks: Array[CsrChunk] = ids.asInstanceOf[IndexedDatasetSpark].matrix.rdd.mapPartitions { it =>
      var rows = new Array[Int](1024)
      var lens = new Array[Int](1024)
      var cols = new Array[Int](16384)
      var nr = 0
      var nc = 0
      while (it.hasNext) {
        val (r, v) = it.next()
        if (nr == rows.length) {
          rows = java.util.Arrays.copyOf(rows, 2 * nr)
          lens = java.util.Arrays.copyOf(lens, 2 * nr)
        }
        val start = nc
        val nz = v.nonZeroes.iterator
        while (nz.hasNext) {
          if (nc == cols.length) {
            require(nc < (1 << 30), "urcco: one Spark partition holds more than 2^30 interactions; repartition the DRM")
            cols = java.util.Arrays.copyOf(cols, 2 * nc)
          }
          cols(nc) = nz.next().index
          nc += 1
        }
        java.util.Arrays.sort(cols, start, nc) // RandomAccessSparseVector iterates in hash order
        rows(nr) = r
        lens(nr) = nc - start
        nr += 1
      }
      Iterator.single(new CsrChunk(java.util.Arrays.copyOf(rows, nr), java.util.Arrays.copyOf(lens, nr), java.util.Arrays.copyOf(cols, nc)))
    }.collect()
    val rp = new Array[Long](nrow + 1)
    for (c <- chunks) {
      var k = 0
      while (k < c.rows.length) {
        require(c.rows(k) >= 0 && c.rows(k) < nrow, s"urcco: DRM row key ${c.rows(k)} outside [0, ${nrow})")
        rp(c.rows(k) + 1) = c.lens(k).toLong
        k += 1
      }
    }
    var i = 0
    while (i < nrow) { rp(i + 1) += rp(i); i += 1 }
    // a Java array holds < 2^31 elements: say so instead of wrapping `.toInt` (config 4's largest matrix has 0.39e9 entries)
    require(rp(nrow) <= Int.MaxValue.toLong - 8,
      s"urcco: ${rp(nrow)} interactions in one event type do not fit one Java int[]; split the event type or raise minEventsPerUser")
    val ci = new Array[Int](rp(nrow).toInt)
    for (c <- chunks) {
      var k = 0
      var off = 0
      while (k < c.rows.length) {
        System.arraycopy(c.cols, off, ci, rp(c.rows(k)).toInt, c.lens(k))
        off += c.lens(k)
        k += 1
      }
    }
    (rp, ci)
  }

  /** Mahout: SimilarityAnalysis.crossOccurrenceDownsampled(datasets, randomSeed) */
  def crossOccurrenceDownsampled(datasets: List[DownsamplableCrossOccurrenceDataset], randomSeed: Int = 0xdeadbeef): List[IndexedDataset] = {
    val sc = datasets.head.iD.matrix.context.asInstanceOf[SparkDistributedContext].sc
    val csr = datasets.map(d => toCsr(d.iD))
    val res = Native.crossOccurrenceDownsampled(
      csr.map(_._1).toArray, csr.map(_._2).toArray, datasets.map(_.iD.matrix.ncol.toLong).toArray,
      datasets.map(_.maxElementsPerRow).toArray, datasets.map(_.maxInterestingElements).toArray,
      datasets.map(_.minLLROpt.getOrElse(Double.NaN)).toArray, randomSeed, 0, numGPUs)
    val a = datasets.head.iD
    val nItemsA = a.matrix.ncol
    datasets.zipWithIndex.map { case (d, i) =>
      val rp = res(3 * i).asInstanceOf[Array[Long]]
      val ci = res(3 * i + 1).asInstanceOf[Array[Int]]
      val llr = res(3 * i + 2).asInstanceOf[Array[Double]]
      val ncol = d.iD.matrix.ncol
      // items without indicators have no DRM row (Mahout's sparse result has none either); rp(nItemsA) < 2^31: the shim
      // refuses results that do not fit Java arrays
      val rows = new scala.collection.mutable.ArrayBuffer[(Int, Vector)]()
      var r = 0
      while (r < nItemsA) {
        if (rp(r + 1) > rp(r)) {
          val v: Vector = new SequentialAccessSparseVector(ncol, (rp(r + 1) - rp(r)).toInt)
          var p = rp(r).toInt
          val e = rp(r + 1).toInt
          while (p < e) { v.setQuick(ci(p), llr(p)); p += 1 }
          rows += ((r, v))
        }
        r += 1
      }
      val drm = drmWrap[Int](sc.parallelize(rows).asInstanceOf[DrmRdd[Int]], nrow = nItemsA, ncol = ncol)
      // the wrap Mahout does: indexedDatasets(0).create(drm, indexedDatasets(0).columnIDs, indexedDatasets(i).columnIDs)
      new IndexedDatasetSpark(drm, a.columnIDs, d.iD.columnIDs).asInstanceOf[IndexedDataset]
    }
  }

  /** Mahout: SimilarityAnalysis.cooccurrencesIDSs(indexedDatasets, randomSeed, maxInterestingItemsPerThing, maxNumInteractions, parOpts).
    * `parOpts` steers Spark partitioning of Mahout's intermediate DRMs; there are none here, it is accepted and ignored. */
  def cooccurrencesIDSs(indexedDatasets: Array[IndexedDataset], randomSeed: Int = 0xdeadbeef, maxInterestingItemsPerThing: Int = 50,
      maxNumInteractions: Int = 500, parOpts: ParOpts = defaultParOpts): List[IndexedDataset] =
    crossOccurrenceDownsampled(
      indexedDatasets.map(new DownsamplableCrossOccurrenceDataset(_, maxNumInteractions, maxInterestingItemsPerThing, None)).toList, randomSeed)
}
