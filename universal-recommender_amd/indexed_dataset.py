"""Host-side mirror of Mahout's IndexedDataset / BiDictionary as the Universal Recommender uses them
(reference src/main/scala/Preparator.scala:111-157, :170-213; package.scala:87-103)."""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np


class BiDictionary:
    """String id <-> dense int index.  Indices follow first appearance in the event stream (decision D8; the
    reference's order is Spark's `distinct().collect()` order, which is arbitrary and unobservable)."""

    def __init__(self, keys: Sequence[str]):
        self._keys: List[str] = list(keys)
        self._index: Dict[str, int] = {k: i for i, k in enumerate(self._keys)}
        if len(self._index) != len(self._keys):
            raise ValueError("BiDictionary keys must be unique")

    @property
    def size(self) -> int:
        return len(self._keys)

    def __len__(self) -> int:
        return len(self._keys)

    def contains(self, key: str) -> bool:
        return key in self._index

    __contains__ = contains

    def get(self, key: str) -> Optional[int]:
        return self._index.get(key)

    def getOrElse(self, key: str, default: int) -> int:
        return self._index.get(key, default)

    def inverse(self, index: int) -> str:
        return self._keys[index]

    @property
    def keys(self) -> List[str]:
        return self._keys


class IndexedDataset:
    """A binary user x item matrix (CSR, values implicit 1.0) with its row and column dictionaries
    (Mahout IndexedDatasetSpark(matrix, rowIDs, columnIDs)).  For the OUTPUT of the CCO build the matrix carries
    LLR values: rows = items of the primary event, columns = items of event i."""

    def __init__(self, row_ptr: np.ndarray, col_idx: np.ndarray, rowIDs: BiDictionary, columnIDs: BiDictionary,
                 values: Optional[np.ndarray] = None):
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
        self.col_idx = np.ascontiguousarray(col_idx, dtype=np.int32)
        self.values = None if values is None else np.ascontiguousarray(values, dtype=np.float64)
        self.rowIDs = rowIDs
        self.columnIDs = columnIDs
        if self.row_ptr.shape != (rowIDs.size + 1,):
            raise ValueError("row_ptr length must be rowIDs.size + 1 (newRowCardinality, Preparator.scala:157,213)")

    @property
    def nrow(self) -> int:
        return self.rowIDs.size

    @property
    def ncol(self) -> int:
        return self.columnIDs.size

    @property
    def nnz(self) -> int:
        return int(self.row_ptr[-1])

    def row(self, i: int):
        s, e = self.row_ptr[i], self.row_ptr[i + 1]
        if self.values is None:
            return list(self.col_idx[s:e].tolist())
        return list(zip(self.col_idx[s:e].tolist(), self.values[s:e].tolist()))

    def create(self, row_ptr, col_idx, rowIDs, columnIDs, values=None) -> "IndexedDataset":
        """IndexedDataset.create(matrix, rowIDs, columnIDs)."""
        return IndexedDataset(row_ptr, col_idx, rowIDs, columnIDs, values)
