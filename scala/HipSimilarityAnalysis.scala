/*
 * HipSimilarityAnalysis -- the Scala host side of liburcco: Mahout-identical signatures for the two calls
 * URAlgorithm.calcAll makes (reference src/main/scala/URAlgorithm.scala:323-329 and :343-346), implemented on the
 * MI355X through the JNI shim jni/urcco_jni.cpp -> include/urcco.h.
 *
 * Drop-in: in URAlgorithm.scala replace `SimilarityAnalysis.` by `HipSimilarityAnalysis.` at the two call sites (see
 * INTEGRATION.md section 4); Preparator.prepare before and `.map(_.asInstanceOf[IndexedDatasetSpark])` / URModel.save after
 * stay untouched.  Cannot be compiled in the build image of this repository (no JDK / Scala / Mahout 0.13.0 jars); the
 * native half is exercised by tests/test_jni_shim.py against a fake JNIEnv.
 */
package com.actionml.urcco

import org.apache.mahout.math.{SequentialAccessSparseVector, Vector}
import org.apache.mahout.math.cf.DownsamplableCrossOccurrenceDataset
import org.apache.mahout.math.indexeddataset.IndexedDataset
import org.apache.mahout.sparkbindings.{DrmRdd, SparkDistributedContext, drmWrap}
import org.apache.mahout.sparkbindings.drm.CheckpointedDrmSpark
import org.apache.mahout.sparkbindings.indexeddataset.IndexedDatasetSpark

import scala.collection.JavaConversions._

/** native methods of liburcco_jni.so (jni/urcco_jni.cpp) */
object Native {
  System.loadLibrary("urcco_jni")

  /** @return Array(3 * n): { rowPtr: Array[Long], colIdx: Array[Int], llr: Array[Double] } per dataset;
    *         throws RuntimeException when the library reports an error (the caller may fall back to Mahout) */
  @native def crossOccurrenceDownsampled(rowPtrs: Array[Array[Long]], colIdxs: Array[Array[Int]], nCols: Array[Long],
      maxElementsPerRow: Array[Int], maxInterestingElements: Array[Int], minLlr: Array[Double] /* NaN = None */,
      seed: Int, device: Int, nGpus: Int /* 0 = every visible GPU */): Array[AnyRef]
  @native def deviceCount(): Int
  @native def shutdown(): Unit
}

object HipSimilarityAnalysis {

  /** engine.json `numGPUs` (additive optional key); 0 = every GPU of the node */
  @volatile var numGPUs: Int = 0

  /** The driver materialises a DRM as CSR: (rowIdx, sorted non-zero column indices).  Values are all 1.0
    * (Preparator.scala:146,205), so only the structure travels; users without a row in this event type get an empty row. */
  private def toCsr(ids: IndexedDataset): (Array[Long], Array[Int]) = {
    val nrow = ids.matrix.nrow.toInt
    val rows = ids.matrix.asInstanceOf[CheckpointedDrmSpark[Int]].rdd
      .map { case (r, v) => (r, v.nonZeroes.map(_.index).toArray.sorted) }.collect()
    val len = new Array[Int](nrow)
    rows.foreach { case (r, c) => len(r) = c.length }
    val rp = new Array[Long](nrow + 1)
    var i = 0
    while (i < nrow) { rp(i + 1) = rp(i) + len(i); i += 1 }
    val ci = new Array[Int](rp(nrow).toInt)
    rows.foreach { case (r, c) => System.arraycopy(c, 0, ci, rp(r).toInt, c.length) }
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
    datasets.zipWithIndex.map { case (d, i) =>
      val rp = res(3 * i).asInstanceOf[Array[Long]]
      val ci = res(3 * i + 1).asInstanceOf[Array[Int]]
      val llr = res(3 * i + 2).asInstanceOf[Array[Double]]
      val ncol = d.iD.matrix.ncol
      // items without indicators have no DRM row (Mahout's sparse result has none either)
      val rows = (0 until a.matrix.ncol).filter(r => rp(r + 1) > rp(r)).map { r =>
        val v: Vector = new SequentialAccessSparseVector(ncol)
        var p = rp(r).toInt
        while (p < rp(r + 1)) { v.setQuick(ci(p), llr(p)); p += 1 }
        r -> v
      }
      val drm = drmWrap[Int](sc.parallelize(rows).asInstanceOf[DrmRdd[Int]], nrow = a.matrix.ncol, ncol = ncol)
      // the wrap Mahout does: indexedDatasets(0).create(drm, indexedDatasets(0).columnIDs, indexedDatasets(i).columnIDs)
      new IndexedDatasetSpark(drm, a.columnIDs, d.iD.columnIDs).asInstanceOf[IndexedDataset]
    }
  }

  /** Mahout: SimilarityAnalysis.cooccurrencesIDSs(indexedDatasets, randomSeed, maxInterestingItemsPerThing, maxNumInteractions) */
  def cooccurrencesIDSs(indexedDatasets: Array[IndexedDataset], randomSeed: Int = 0xdeadbeef, maxInterestingItemsPerThing: Int = 50,
      maxNumInteractions: Int = 500): List[IndexedDataset] =
    crossOccurrenceDownsampled(
      indexedDatasets.map(new DownsamplableCrossOccurrenceDataset(_, maxNumInteractions, maxInterestingItemsPerThing, None)).toList, randomSeed)
}
