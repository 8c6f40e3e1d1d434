"""Host-side mirror of Preparator.prepare (reference src/main/scala/Preparator.scala:44-87 and the two
`object IndexedDatasetSpark.apply` overloads at :102-158, :160-214): one binary user x item matrix per event type,
all sharing one user dictionary; `minEventsPerUser` filters users on their RAW primary-event count.
`Preparator.prepare` builds them on the host (numpy); `Preparator.prepare_on_device` builds them with the GPU ingest
kernels (ingest.py: device dictionaries + CSR builder) from 64-bit hashes of the id strings -- same result."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .data_source import TrainingData
from .indexed_dataset import BiDictionary, IndexedDataset


@dataclass
class PreparedData:
    """Preparator.scala:91-93."""
    actions: List[Tuple[str, IndexedDataset]]
    fields: Dict[str, Dict[str, object]]


def _first_appearance(keys: Sequence[str]) -> List[str]:
    return list(dict.fromkeys(keys))


def _indexed_dataset(elements: Sequence[Tuple[str, str]], existing_row_ids: Optional[BiDictionary]) -> IndexedDataset:
    """IndexedDatasetSpark.apply(elements, existingRowIDs) Preparator.scala:160-214."""
    if existing_row_ids is None:
        row_ids = BiDictionary(_first_appearance([u for u, _ in elements]))            # :170
        filtered = elements
    else:
        row_ids = existing_row_ids                                                      # :173-179, never extended
        filtered = [(u, i) for (u, i) in elements if u in row_ids]
    column_ids = BiDictionary(_first_appearance([i for _, i in filtered]))              # :184-186
    n_rows, n_cols = row_ids.size, column_ids.size
    if filtered:
        r = np.fromiter((row_ids.get(u) for u, _ in filtered), np.int64, count=len(filtered))
        c = np.fromiter((column_ids.get(i) for _, i in filtered), np.int64, count=len(filtered))
        key = np.unique(r * max(n_cols, 1) + c)        # setQuick(col, 1.0): duplicates collapse (:205); sorted by (row, col)
        rr = key // max(n_cols, 1)
        cc = (key - rr * max(n_cols, 1)).astype(np.int32)
    else:
        rr = np.zeros(0, np.int64)
        cc = np.zeros(0, np.int32)
    row_ptr = np.zeros(n_rows + 1, np.int64)
    np.cumsum(np.bincount(rr, minlength=n_rows), out=row_ptr[1:])
    return IndexedDataset(row_ptr, cc, row_ids, column_ids)


def _min_events_row_ids(elements: Sequence[Tuple[str, str]], min_events: int) -> BiDictionary:
    """IndexedDatasetSpark.apply(elements, minEventsPerUser): only its rowIDs are used (Preparator.scala:57-62).
    groupByKey ... items.size counts RAW events, duplicates included (:129-132)."""
    counts: Dict[str, int] = {}
    for u, _ in elements:
        counts[u] = counts.get(u, 0) + 1
    return BiDictionary([u for u, c in counts.items() if c >= min_events])


HASH_SEED = 0
CHECK_SEED = 0x9E3779B97F4A7C15      # second, independent hash of the same string: the collision check


def hash_keys(strings: Sequence[str], seed: int = HASH_SEED, library=None) -> np.ndarray:
    """64-bit keys of id strings: XXH64(seed) of the UTF-8 bytes, evaluated by the library's host helper
    (urcco_hash_strings: native, multi-threaded) -- a fixed function of the string, so every rank of a multi-GPU build maps
    the same id to the same key.  ~0 is the device dictionary's reserved value and is remapped to 0.  Two distinct strings
    collide with probability ~n^2 / 2^65 (3e-5 for a billion ids); `prepare_on_device` detects that with a second hash."""
    import ctypes as C
    from . import _lib
    lib = library if library is not None else _lib.lib()
    enc = [x.encode("utf-8") for x in strings]
    n = len(enc)
    offsets = np.zeros(n + 1, np.int64)
    if n:
        np.cumsum(np.fromiter(map(len, enc), np.int64, count=n), out=offsets[1:])
    blob = b"".join(enc)
    out = np.empty(max(n, 1), np.uint64)
    buf = np.frombuffer(blob, np.uint8) if blob else np.zeros(1, np.uint8)
    _lib.check(lib.urcco_hash_strings(buf.ctypes.data, offsets.ctypes.data, n, C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), out.ctypes.data), lib)
    return out[:n].view(np.int64)


class Preparator:
    def prepare_on_device(self, trainingData: TrainingData, sess, keep_on_device: bool = False):
        """Preparator.prepare through the GPU ingest kernels.  The host hashes the id strings, the device builds the
        dictionaries and the matrices (ingest.prepare_device), and the BiDictionaries are rebuilt from the stream positions
        of every id's first occurrence.  Returns PreparedData (matrices copied to the host); with keep_on_device also the
        ingest.DevPreparedData whose matrices can go straight into the CCO build without touching PCIe again."""
        import torch
        from . import ingest
        if not trainingData.actions:
            raise ValueError("no event type with events")
        dev_actions = []
        for name, elements in trainingData.actions:
            users, items = [u for u, _ in elements], [i for _, i in elements]
            keys = [torch.from_numpy(hash_keys(x, sd, sess.lib)).to(sess.device) for x in (users, items) for sd in (HASH_SEED, CHECK_SEED)]
            dev_actions.append((name, keys[0], keys[2], keys[1], keys[3]))      # (user keys, item keys, user check keys, item check keys)
        dp = ingest.prepare_device(sess, dev_actions, trainingData.minEventsPerUser)   # raises on a hash collision
        primary = trainingData.actions[0][1]
        row_ids = BiDictionary([primary[p][0] for p in dp.user_first_pos.cpu().numpy()])
        out: List[Tuple[str, IndexedDataset]] = []
        for (name, elements), ev in zip(trainingData.actions, dp.events):
            column_ids = BiDictionary([elements[p][1] for p in ev.item_first_pos.cpu().numpy()])
            rp = ev.matrix.row_ptr.cpu().numpy()
            ci = ev.matrix.col_idx[: int(rp[-1])].cpu().numpy()
            out.append((name, IndexedDataset(rp, ci, row_ids, column_ids)))
        pd = PreparedData(out, trainingData.fields)
        return (pd, dp) if keep_on_device else pd

    def prepare(self, trainingData: TrainingData) -> PreparedData:
        user_dictionary: Optional[BiDictionary] = None
        out: List[Tuple[str, IndexedDataset]] = []
        for pos, (event_name, elements) in enumerate(trainingData.actions):
            if pos == 0 and trainingData.minEventsPerUser is not None:
                d_row_ids = _min_events_row_ids(elements, trainingData.minEventsPerUser)     # :57
                ids = _indexed_dataset(elements, d_row_ids)                                  # :62
            else:
                ids = _indexed_dataset(elements, user_dictionary)                            # :71
            user_dictionary = ids.rowIDs                                                     # :63, :72
            out.append((event_name, ids))
        return PreparedData(out, trainingData.fields)
