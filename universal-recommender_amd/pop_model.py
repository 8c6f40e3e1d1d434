"""Host-side mirror of PopModel (reference src/main/scala/PopModel.scala:55-179) and of URAlgorithm.getRanksRDD
(src/main/scala/URAlgorithm.scala:537-560) on top of the device interval histogram `urcco_dev_pop_counts`.

The reference counts each interval with its own PEventStore.find + groupByKey; here the event stream (item ids, event
times) is uploaded once per ranking and ONE kernel pass fills the histograms of every interval the ranking type needs
(popular 1, trending 2, hot 3).  The joins that follow are per-item arithmetic on the three count vectors.
`random` ranks are `Random.nextDouble` per item in the reference (not reproducible there either); `userDefined` is empty.
No CPU fallback: needs the HIP library and a device (or the test-only simulator build)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .indexed_dataset import BiDictionary


class RankingFieldName:
    UserRank, UniqueRank, PopRank, TrendRank, HotRank, UnknownRank = "userRank", "uniqueRank", "popRank", "trendRank", "hotRank", "unknownRank"

    @staticmethod
    def toSeq():   # PopModel.scala:39
        return [RankingFieldName.UserRank, RankingFieldName.UniqueRank, RankingFieldName.PopRank, RankingFieldName.TrendRank, RankingFieldName.HotRank]


class RankingType:
    Popular, Trending, Hot, UserDefined, Random = "popular", "trending", "hot", "userDefined", "random"


nameByType = {RankingType.Popular: RankingFieldName.PopRank, RankingType.Trending: RankingFieldName.TrendRank, RankingType.Hot: RankingFieldName.HotRank,
              RankingType.UserDefined: RankingFieldName.UserRank, RankingType.Random: RankingFieldName.UniqueRank}


class PopModel:
    """events = (event name, target item id or None, time in ms since the epoch) in stream order; fields = item -> properties
    (the reference's fieldsRDD, only used by `random`)."""

    def __init__(self, events: Sequence[Tuple[str, Optional[str], int]], fields: Optional[Dict[str, dict]], sess):
        self.sess = sess
        self.fields = fields or {}
        self.names = [e[0] for e in events]
        items: List[str] = []
        seen = {}
        ids = np.empty(len(events), np.int32)
        for p, (_, item, _) in enumerate(events):
            if item is None:
                ids[p] = -1
                continue
            i = seen.get(item)
            if i is None:
                i = seen[item] = len(items)
                items.append(item)
            ids[p] = i
        self.itemIDs = BiDictionary(items)
        self._ids = ids
        self._times = np.asarray([e[2] for e in events], np.int64)
        dev = sess.device
        self._d_times = torch.from_numpy(self._times.copy()).to(dev) if len(events) else torch.zeros(1, dtype=torch.int64, device=dev)

    def _counts(self, event_names: Sequence[str], bounds: Sequence[int]) -> np.ndarray:
        """counts[b][i] through the device histogram (one pass for all intervals)."""
        n_int = len(bounds) - 1
        n_items = self.itemIDs.size
        if n_items == 0:
            return np.zeros((n_int, 0), np.int32)
        wanted = set(event_names)   # empty = every event name (PopModel.scala:194: `if (eventNames.nonEmpty) Some(eventNames) else None`)
        sel = np.fromiter((not wanted or n in wanted for n in self.names), bool, count=len(self.names))
        ids = np.where(sel, self._ids, -1).astype(np.int32)
        dev = self.sess.device
        d_ids = torch.from_numpy(ids).to(dev) if ids.size else torch.zeros(1, dtype=torch.int32, device=dev)
        counts = torch.empty(n_int * n_items, dtype=torch.int32, device=dev)
        b = (C.c_int64 * (n_int + 1))(*[int(x) for x in bounds])
        _lib.check(self.sess.lib.urcco_dev_pop_counts(self.sess.handle, ids.size, d_ids.data_ptr(), self._d_times.data_ptr(), n_items, n_int, b,
                                                      counts.data_ptr()), self.sess.lib)
        self.sess.synchronize()
        return counts.cpu().numpy().reshape(n_int, n_items)

    def calc(self, modelName: str, eventNames: Sequence[str], duration: int = 0, end_ms: Optional[int] = None,
             now_ms: Optional[int] = None, seed: int = 0) -> Dict[str, float]:
        """PopModel.calc :59-97.  duration in seconds; end_ms = parsed offsetDate (None = now_ms, None = the wall clock: the
        reference's `DateTime.now` :66-74)."""
        if end_ms is None and now_ms is None:
            import time
            now_ms = int(time.time() * 1000)
        end = int(end_ms if end_ms is not None else now_ms)
        start = end - int(duration) * 1000
        inv = self.itemIDs.inverse
        if modelName == RankingType.Popular:                                      # calcPopular :113-122
            c = self._counts(eventNames, [start, end])[0]
            return {inv(int(i)): float(c[i]) for i in np.nonzero(c)[0]}
        if modelName == RankingType.Trending:                                     # calcTrending :128-147
            half = (end - start) // 2
            older, newer = self._counts(eventNames, [start, start + half, end])
            if not older.any():
                return {}
            both = np.nonzero((older > 0) & (newer > 0))[0]
            return {inv(int(i)): float(int(newer[i]) - int(older[i])) for i in both}
        if modelName == RankingType.Hot:                                          # calcHot :152-179
            third = (end - start) // 3
            older, middle, newer = self._counts(eventNames, [start, start + third, start + 2 * third, end])
            if not older.any() or not middle.any():
                return {}
            all3 = np.nonzero((older > 0) & (middle > 0) & (newer > 0))[0]
            return {inv(int(i)): float((int(newer[i]) - int(middle[i])) - (int(middle[i]) - int(older[i]))) for i in all3}
        if modelName == RankingType.Random:                                       # calcRandom :100-110 (Random.nextDouble per item)
            rng = np.random.default_rng(seed)
            keys = list(dict.fromkeys([inv(int(i)) for i in np.unique(self._ids[(self._ids >= 0) & (self._times >= start) & (self._times < end)])]
                                      + list(self.fields)))
            return {k: float(rng.random()) for k in keys}
        return {}                                                                 # userDefined / unknown: sc.emptyRDD


def getRanks(rankings: Sequence[dict], popModel: PopModel, modelEventNames: Sequence[str], now_ms: int) -> Dict[str, Dict[str, float]]:
    """URAlgorithm.getRanksRDD :537-560: one PopModel.calc per `rankings` entry, folded by full outer joins into
    item -> {ranking field name: rank}.  rankings entries: {name?, type?, eventNames?, duration_s?, end_ms?}."""
    out: Dict[str, Dict[str, float]] = {}
    for r in rankings:
        rtype = r.get("type") or RankingType.Popular
        field = r.get("name") or nameByType.get(rtype, RankingFieldName.UnknownRank)
        names = r["eventNames"] if r.get("eventNames") is not None else list(modelEventNames[:1])   # Option.getOrElse: Some(Seq()) stays empty = all events
        ranks = popModel.calc(rtype, names, int(r.get("duration_s", 3650 * 86400)), r.get("end_ms"), now_ms)
        for item, v in ranks.items():
            out.setdefault(item, {})[field] = v
    return out


def propertiesWithRanks(fields: Dict[str, dict], ranks: Dict[str, Dict[str, float]]) -> Dict[str, dict]:
    """calcAll's propertiesRDD (URAlgorithm.scala:351-358): fields fullOuterJoin ranks, `fieldsPropMap ++ rankPropMap`."""
    out = {}
    for item in list(fields) + [i for i in ranks if i not in fields]:
        m = dict(fields.get(item, {}))
        m.update(ranks.get(item, {}))
        out[item] = m
    return out
