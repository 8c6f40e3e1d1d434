"""Device-side Preparator: `Preparator.prepare` (reference src/main/scala/Preparator.scala:44-87 and the two
IndexedDatasetSpark builders at :102-158, :160-214) on the GPU, from event streams of 64-bit keys.

The host keeps the id strings; it hands the device one pair of key arrays (user key, item key) per event type -- a 64-bit
hash of each string, or integer ids -- and gets back, per event type, the binary user x item CSR matrix in HBM plus, for
every dictionary, the stream position of each id's first occurrence (which is all it needs to rebuild the
id -> string BiDictionary: dense ids follow first appearance, decision D8 of DESIGN.md).  Semantics mirrored exactly:
  * the primary event type defines the ONE user dictionary; with `minEventsPerUser` only users with at least that many
    RAW primary events (duplicates included, :129-132) are kept, and the primary is rebuilt restricted to them so
    that items only dropped users touched vanish from its column dictionary (:57-63);
  * secondary event types drop events of users that are not in the dictionary, which is never extended (:173-179);
  * column dictionaries are per event type, over the events that survive the user filter (:184-186);
  * duplicates collapse (`setQuick(col, 1.0)`, :146, :205); all matrices have nrow = size(userDictionary) (:157, :213).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch

from .device import DevCsr, DeviceSession, _ptr


class DevDictionary:
    """urcco_key_table + the first-occurrence positions of its ids (first_pos[id] = index into the stream it was built from)."""

    def __init__(self, sess: DeviceSession, handle: C.c_void_p, n_ids: int, first_pos: torch.Tensor):
        self._sess = sess
        self.handle = handle
        self.n_ids = int(n_ids)
        self.first_pos = first_pos[: self.n_ids]

    def close(self):
        if self.handle:
            self._sess.lib.urcco_key_table_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def dictionary_build(sess: DeviceSession, keys: torch.Tensor, select: Optional[torch.Tensor] = None, min_count: int = 1) -> DevDictionary:
    """keys: uint64 stream as an int64 tensor (bit pattern); select: int32[n], positions with a negative entry are ignored."""
    n = keys.numel()
    first_pos = sess.empty(max(n, 1), torch.int64)
    handle, n_ids = C.c_void_p(), C.c_int64()
    sess._check(sess.lib.urcco_dev_dictionary_build(sess.handle, n, _ptr(keys), _ptr(select), int(min_count), _ptr(first_pos), C.byref(handle),
                                                    C.byref(n_ids)))
    return DevDictionary(sess, handle, n_ids.value, first_pos)


def dictionary_lookup(sess: DeviceSession, d: DevDictionary, keys: torch.Tensor, select: Optional[torch.Tensor] = None) -> torch.Tensor:
    ids = sess.empty(max(keys.numel(), 1), torch.int32)
    sess._check(sess.lib.urcco_dev_dictionary_lookup(sess.handle, d.handle, keys.numel(), _ptr(keys), _ptr(select), _ptr(ids)))
    return ids[: keys.numel()]


class HashCollision(RuntimeError):
    """Two different id strings share a 64-bit key (detected through the second hash); re-key with another seed."""


def dictionary_verify(sess: DeviceSession, d: DevDictionary, keys: torch.Tensor, check_keys: torch.Tensor, select: Optional[torch.Tensor] = None) -> int:
    """Number of stream positions whose check key differs from the check key of their id's first occurrence (0 = no collision)."""
    bad = C.c_int64()
    sess._check(sess.lib.urcco_dev_dictionary_verify(sess.handle, d.handle, keys.numel(), _ptr(keys), _ptr(select), _ptr(check_keys),
                                                     _ptr(d.first_pos) if d.n_ids else _ptr(sess.empty(1, torch.int64)), C.byref(bad)))
    return int(bad.value)


def dictionary_verify_against(sess: DeviceSession, d: DevDictionary, keys: torch.Tensor, check_keys: torch.Tensor, dict_check_keys: torch.Tensor) -> int:
    """dictionary_verify for another stream looked up in `d`: its check keys against the check keys of the ids' first occurrences
    in the stream the dictionary was built from."""
    bad = C.c_int64()
    sess._check(sess.lib.urcco_dev_dictionary_verify_against(sess.handle, d.handle, keys.numel(), _ptr(keys), None, _ptr(check_keys), _ptr(dict_check_keys),
                                                             _ptr(d.first_pos) if d.n_ids else _ptr(sess.empty(1, torch.int64)), C.byref(bad)))
    return int(bad.value)


def csr_from_pairs(sess: DeviceSession, rows: torch.Tensor, cols: torch.Tensor, n_rows: int, n_cols: int) -> DevCsr:
    n = rows.numel()
    out_rp = sess.empty(n_rows + 1, torch.int64)
    out_ci = sess.empty(max(n, 1), torch.int32)
    nnz = C.c_int64()
    sess._check(sess.lib.urcco_dev_csr_from_pairs(sess.handle, n, _ptr(rows), _ptr(cols), n_rows, _ptr(out_rp), _ptr(out_ci), C.byref(nnz)))
    return DevCsr(n_rows, n_cols, out_rp, out_ci, int(nnz.value))


@dataclass
class DevPreparedEvent:
    name: str
    matrix: DevCsr                 # nrow = size of the user dictionary, ncol = size of this event type's item dictionary
    item_first_pos: torch.Tensor   # int64[ncol]: index into THIS event type's stream of each item id's first (surviving) occurrence


@dataclass
class DevPreparedData:
    user_first_pos: torch.Tensor   # int64[n_users]: index into the PRIMARY stream of each user id's first occurrence
    events: List[DevPreparedEvent]


def prepare_device(sess: DeviceSession, actions: Sequence[Tuple[str, torch.Tensor, torch.Tensor]],
                   min_events_per_user: Optional[int] = None) -> DevPreparedData:
    """Preparator.prepare.  actions[d] = (event name, user keys, item keys[, user check keys, item check keys]) with the keys
    as int64 tensors (uint64 bit patterns) resident on the session's device; actions[0] is the primary event type.  With
    check keys (a second, independent hash of the same strings) every dictionary is verified: HashCollision if two
    different strings were merged into one id."""
    if not actions:
        raise ValueError("need at least the primary event type")
    pu = actions[0][1]
    users = dictionary_build(sess, pu, None, min_events_per_user if min_events_per_user is not None else 1)
    try:
        n_users = users.n_ids
        user_first = users.first_pos.clone()
        if len(actions[0]) >= 5 and dictionary_verify(sess, users, pu, actions[0][3]):
            raise HashCollision("two user ids share a 64-bit key")
        out: List[DevPreparedEvent] = []
        for act in actions:
            name, uk, ik = act[0], act[1], act[2]
            if uk.numel() != ik.numel():
                raise ValueError(f"event type {name}: user and item key streams differ in length")
            if act is not actions[0] and len(act) >= 5 and len(actions[0]) >= 5 and dictionary_verify_against(sess, users, uk, act[3], actions[0][3]):
                raise HashCollision(f"event type {name}: a user id shares its 64-bit key with a different user id of the primary event type")
            rows = dictionary_lookup(sess, users, uk)                      # -1: user not in the dictionary -> event dropped
            items = dictionary_build(sess, ik, rows, 1)                     # column ids over the surviving events only
            try:
                if len(act) >= 5 and dictionary_verify(sess, items, ik, act[4], rows):
                    raise HashCollision(f"event type {name}: two item ids share a 64-bit key")
                cols = dictionary_lookup(sess, items, ik, rows)
                m = csr_from_pairs(sess, rows, cols, n_users, items.n_ids)
                out.append(DevPreparedEvent(name, m, items.first_pos.clone()))
            finally:
                items.close()
        return DevPreparedData(user_first, out)
    finally:
        users.close()
