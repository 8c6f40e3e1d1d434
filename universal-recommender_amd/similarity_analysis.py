"""Host-side mirror of the two Mahout entry points the Universal Recommender calls (reference
src/main/scala/URAlgorithm.scala:323-329, :343-346) with Mahout-identical names, argument meaning and defaults:

    SimilarityAnalysis.cooccurrencesIDSs(indexedDatasets, randomSeed, maxInterestingItemsPerThing, maxNumInteractions)
    SimilarityAnalysis.crossOccurrenceDownsampled(datasets, randomSeed)

Both marshal the CSR matrices across the C ABI (include/urcco.h, host level) exactly as the JNI shim of INTEGRATION.md
would and wrap the returned indicator matrices as IndexedDataset(rowIDs = A.columnIDs, columnIDs = B_i.columnIDs).
No CPU fallback: the HIP library and a device are required."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from .indexed_dataset import IndexedDataset


@dataclass
class DownsamplableCrossOccurrenceDataset:
    """Mahout DownsamplableCrossOccurrenceDataset(iD, maxElementsPerRow = 500, maxInterestingElements = 50, minLLROpt = None)."""
    iD: IndexedDataset
    maxElementsPerRow: int = 500
    maxInterestingElements: int = 50
    minLLROpt: Optional[float] = None


@dataclass
class DatasetStats:
    nnz_raw: int
    nnz_sampled: int
    pairs: int
    nnz_out: int
    rows_by_bin: List[int]
    ms_total: float


last_stats: List[DatasetStats] = []


def _seed_to_int(seed: int) -> int:
    s = int(seed) & 0xFFFFFFFF
    return s - (1 << 32) if s >= (1 << 31) else s


def crossOccurrenceDownsampled(datasets: Sequence[DownsamplableCrossOccurrenceDataset], randomSeed: int = 0xdeadbeef,
                               device: int = 0, rowRateMode: int = _lib.ROW_RATE_MAHOUT_INT_DIV, library=None, numGPUs: int = 1) -> List[IndexedDataset]:
    """numGPUs: GPUs of this process to use (engine.json `numGPUs`; 0 = every visible one, collectives through RCCL)."""
    global last_stats
    if len(datasets) == 0:
        raise ValueError("crossOccurrenceDownsampled needs at least the primary dataset")
    lib = library if library is not None else _lib.lib()
    n = len(datasets)
    arr = (_lib.Dataset * n)()
    keep = []
    for d, ds in enumerate(datasets):
        m = ds.iD
        rp = np.ascontiguousarray(m.row_ptr, np.int64)
        ci = np.ascontiguousarray(m.col_idx, np.int32)
        keep += [rp, ci]
        arr[d].matrix.n_rows = m.nrow
        arr[d].matrix.n_cols = m.ncol
        arr[d].matrix.row_ptr = rp.ctypes.data
        arr[d].matrix.col_idx = ci.ctypes.data if ci.size else None
        arr[d].max_elements_per_row = int(ds.maxElementsPerRow)
        arr[d].max_interesting_elements = int(ds.maxInterestingElements)
        arr[d].has_min_llr = int(ds.minLLROpt is not None)
        arr[d].min_llr = float(ds.minLLROpt) if ds.minLLROpt is not None else 0.0
    opts = _lib.Options(device=device, row_rate_mode=rowRateMode, n_gpus=numGPUs)
    out = (_lib.Indicators * n)()
    stats = (_lib.DatasetStats * n)()
    _lib.check(lib.urcco_cross_occurrence_downsampled(arr, n, _seed_to_int(randomSeed), C.byref(opts), out, stats), lib)
    try:
        primary = datasets[0].iD
        result: List[IndexedDataset] = []
        last_stats = []
        for d in range(n):
            o = out[d]
            rp = np.ctypeslib.as_array(o.row_ptr, shape=(o.n_rows + 1,)).copy()
            nnz = int(o.nnz)
            ci = np.ctypeslib.as_array(o.col_idx, shape=(max(nnz, 1),))[:nnz].copy()
            llr = np.ctypeslib.as_array(o.llr, shape=(max(nnz, 1),))[:nnz].copy()
            # indexedDatasets(0).create(drm, indexedDatasets(0).columnIDs, indexedDatasets(i).columnIDs)
            result.append(primary.create(rp, ci, primary.columnIDs, datasets[d].iD.columnIDs, llr))
            st = stats[d]
            last_stats.append(DatasetStats(st.nnz_raw, st.nnz_sampled, st.pairs, st.nnz_out, list(st.rows_by_bin), st.ms_total))
        return result
    finally:
        lib.urcco_free_indicators(out, n)


def cooccurrencesIDSs(indexedDatasets: Sequence[IndexedDataset], randomSeed: int = 0xdeadbeef, maxInterestingItemsPerThing: int = 50,
                      maxNumInteractions: int = 500, device: int = 0, rowRateMode: int = _lib.ROW_RATE_MAHOUT_INT_DIV,
                      library=None, numGPUs: int = 1) -> List[IndexedDataset]:
    ds = [DownsamplableCrossOccurrenceDataset(d, maxNumInteractions, maxInterestingItemsPerThing, None) for d in indexedDatasets]
    return crossOccurrenceDownsampled(ds, randomSeed, device, rowRateMode, library, numGPUs)
