"""Host-side mirror of the train side of URAlgorithm (reference src/main/scala/URAlgorithm.scala:142-171 params,
:195-247 defaults, :292-369 train/calcAll) and of IndexedDatasetConversions.toStringMapRDD (package.scala:82-110).

calcAll builds what the reference hands to URModel.save: the per-event indicator matrices (the CCO build on the GPU) and --
`calcPopular` -- the item properties joined with the PopModel ranks (getRanksRDD :537-560, device interval histogram behind
pop_model.py).  Elasticsearch and the query side are out of scope."""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

from . import similarity_analysis as SimilarityAnalysis
from .indexed_dataset import IndexedDataset
from .preparator import PreparedData


class DefaultURAlgoParams:
    """URAlgorithm.scala:53-70."""
    MaxEventsPerEventType = 500
    MaxCorrelatorsPerEventType = 50
    RecsModel = "all"
    BackfillFieldName = "popRank"      # RankingFieldName.PopRank :64
    BackfillType = "popular"           # RankingType.Popular :65
    BackfillDuration = "3650 days"     # :66


@dataclass
class RankingParams:
    """URAlgorithm.scala:110-117."""
    name: Optional[str] = None
    type: Optional[str] = None
    eventNames: Optional[List[str]] = None
    offsetDate: Optional[str] = None
    endDate: Optional[str] = None
    duration: Optional[str] = None


# scala.concurrent.duration.Duration's unit words (Duration.scala: timeUnitLabels -- every label, its plural, and the short forms)
_NS = {"d": 86400 * 10**9, "day": 86400 * 10**9, "h": 3600 * 10**9, "hr": 3600 * 10**9, "hour": 3600 * 10**9, "m": 60 * 10**9, "min": 60 * 10**9, "minute": 60 * 10**9,
       "s": 10**9, "sec": 10**9, "second": 10**9, "ms": 10**6, "milli": 10**6, "millis": 10**6, "millisecond": 10**6,
       "µs": 10**3, "micro": 10**3, "micros": 10**3, "microsecond": 10**3, "ns": 1, "nano": 1, "nanos": 1, "nanosecond": 1}
_DURATION_NS = dict(_NS)
_DURATION_NS.update({k + "s": v for k, v in _NS.items() if len(k) > 2 and not k.endswith("s")})   # "days", "hours", "seconds", "milliseconds", ...


def duration_seconds(text: str) -> int:
    """scala.concurrent.duration.Duration("3650 days").toSeconds.toInt (URAlgorithm.scala:542-543): number and unit with or
    without a blank between them, every unit Duration accepts (d/day .. ns/nanosecond, plurals), truncation towards zero by
    toSeconds and the 32-bit wrap of Long.toInt."""
    t = text.strip()
    i = 0
    while i < len(t) and (t[i].isdigit() or t[i] in "+-.eE"):
        i += 1
    num, unit = t[:i], t[i:].strip().lower()
    if not num or unit not in _DURATION_NS:
        raise ValueError(f"bad duration {text!r}")
    try:
        ns = int(num) * _DURATION_NS[unit]
    except ValueError:
        ns = int(float(num) * _DURATION_NS[unit])
    secs = abs(ns) // 10**9 * (1 if ns >= 0 else -1)
    return (secs + 2**31) % 2**32 - 2**31


_ISO_RE = None


def _iso_ms(text: Optional[str]) -> Optional[int]:
    """ISODateTimeFormat.dateTimeParser().parseDateTime (PopModel.scala:66-74): date, optional 'T' time with any number of
    fraction digits, optional offset ('Z', +hh:mm, +hhmm, +hh); extended and basic (yyyyMMdd'T'HHmmss) forms.  A string without
    an offset is read in the DEFAULT time zone, as Joda does (the JVM's user.timezone == this process's local zone).  A bad
    date falls back to `now` in the reference, after a warning: None (and the same warning) here."""
    global _ISO_RE
    if not text:
        return None
    import re
    from datetime import datetime, timedelta, timezone
    if _ISO_RE is None:
        _ISO_RE = re.compile(r"^(\d{4})-?(\d{2})?-?(\d{2})?(?:T(\d{2})?:?(\d{2})?:?(\d{2})?(?:[.,](\d+))?)?(Z|[+-]\d{2}(?::?\d{2})?)?$")
    m = _ISO_RE.match(text.strip())
    try:
        if not m:
            raise ValueError(text)
        y, mo, da, hh, mi, ss, frac, off = m.groups()
        d = datetime(int(y), int(mo or 1), int(da or 1), int(hh or 0), int(mi or 0), int(ss or 0))
        ms = int((frac + "000")[:3]) if frac else 0
        if off is None:
            base = d.astimezone()            # naive -> the process's local zone (Joda: DateTimeZone.getDefault)
        elif off == "Z":
            base = d.replace(tzinfo=timezone.utc)
        else:
            sign = -1 if off[0] == "-" else 1
            digits = off[1:].replace(":", "")
            base = d.replace(tzinfo=timezone(sign * timedelta(hours=int(digits[:2]), minutes=int(digits[2:4] or 0))))
        return int(base.timestamp()) * 1000 + ms
    except (ValueError, OverflowError):
        import logging
        logging.getLogger("universal_recommender_amd").warning("bad date %r: falling back to now (PopModel.scala:70)", text)
        return None


@dataclass
class IndicatorParams:
    """URAlgorithm.scala:136-140."""
    name: str
    maxItemsPerUser: Optional[int] = None
    maxCorrelatorsPerItem: Optional[int] = None
    minLLR: Optional[float] = None


@dataclass
class URAlgorithmParams:
    """The hot-path subset of URAlgorithm.scala:142-171 (other keys are accepted and ignored)."""
    appName: str = ""
    indexName: str = ""
    typeName: str = ""
    recsModel: Optional[str] = None
    eventNames: Optional[List[str]] = None
    maxEventsPerEventType: Optional[int] = None
    maxCorrelatorsPerEventType: Optional[int] = None
    indicators: Optional[List[IndicatorParams]] = None
    seed: Optional[int] = None
    rankings: Optional[List[RankingParams]] = None      # :159
    availableDateName: Optional[str] = None   # :160-162: item properties that hold dates (URModel.extractJvalue turns them into dates)
    expireDateName: Optional[str] = None
    dateName: Optional[str] = None
    numGPUs: Optional[int] = None      # additive key: GPUs of the node the CCO build may use; absent = 1, 0 = every visible one
    ccoBackend: Optional[str] = None   # additive key: "hip" (default) or "mahout" (the host falls back to the reference path: not available here)

    @staticmethod
    def from_engine_json(engine: dict, name: str = "ur") -> "URAlgorithmParams":
        algos = [a for a in engine["algorithms"] if a.get("name") == name]
        if not algos:
            raise ValueError(f"no algorithm named {name!r} in engine.json")
        p = algos[0]["params"]
        inds = p.get("indicators")
        rk = p.get("rankings")
        return URAlgorithmParams(
            rankings=None if rk is None else [RankingParams(r.get("name"), r.get("type"), r.get("eventNames"), r.get("offsetDate"), r.get("endDate"),
                                                            r.get("duration")) for r in rk],
            numGPUs=p.get("numGPUs"), ccoBackend=p.get("ccoBackend"),
            availableDateName=p.get("availableDateName"), expireDateName=p.get("expireDateName"), dateName=p.get("dateName"),
            appName=p.get("appName", ""), indexName=p.get("indexName", ""), typeName=p.get("typeName", ""),
            recsModel=p.get("recsModel"), eventNames=p.get("eventNames"),
            maxEventsPerEventType=p.get("maxEventsPerEventType"), maxCorrelatorsPerEventType=p.get("maxCorrelatorsPerEventType"),
            indicators=None if inds is None else [IndicatorParams(i["name"], i.get("maxItemsPerUser"), i.get("maxCorrelatorsPerItem"),
                                                                  i.get("minLLR")) for i in inds],
            seed=p.get("seed"))


def _get_or_else(v, default):
    """Scala `Option.getOrElse`: only an ABSENT value takes the default -- an explicit 0 is passed through (and rejected by the
    library as BAD_ARG, where Mahout would misbehave), it is not silently replaced."""
    return default if v is None else v


class URAlgorithm:
    def __init__(self, ap: URAlgorithmParams, device: int = 0, library=None, eventStore=None, sess=None):
        """eventStore: the timed event stream PopModel reads from PEventStore in the reference -- (event name, target item id or
        None, time in ms) in stream order; None = no events (every ranking is empty).  sess: DeviceSession for the PopModel
        histograms (created on `device` when needed)."""
        self.ap = ap
        self.device = device
        self.library = library
        self.eventStore = eventStore
        self.sess = sess
        self.recsModel = ap.recsModel or DefaultURAlgoParams.RecsModel
        if not ap.eventNames and not ap.indicators:                                            # :224-226
            raise ValueError("Must have either \"eventNames\" or \"indicators\" in algorithm parameters.")
        self.modelEventNames = [i.name for i in ap.indicators] if ap.indicators else list(ap.eventNames)  # :230-234
        if ap.ccoBackend not in (None, "hip"):
            raise ValueError(f"ccoBackend={ap.ccoBackend!r}: only the HIP backend exists in this package (\"mahout\" is the reference's own path)")
        if ap.numGPUs is not None and ap.numGPUs < 0:
            raise ValueError("numGPUs must be >= 0 (0 = every visible GPU)")
        self.numGPUs = 1 if ap.numGPUs is None else int(ap.numGPUs)

    @property
    def dateNames(self) -> List[str]:
        """:264-267: Seq(dateName, availableDateName, expireDateName).flatten.distinct -- what calcAll hands to URModel.save."""
        out: List[str] = []
        for n in (self.ap.dateName, self.ap.availableDateName, self.ap.expireDateName):
            if n is not None and n not in out:
                out.append(n)
        return out

    @property
    def rankingsParams(self) -> List[RankingParams]:
        """:250-256: the default is one all-time popularity ranking on the primary event; one entry per ranking type."""
        rk = self.ap.rankings if self.ap.rankings is not None else [RankingParams(
            DefaultURAlgoParams.BackfillFieldName, DefaultURAlgoParams.BackfillType, self.modelEventNames[:1], None, None, DefaultURAlgoParams.BackfillDuration)]
        by_type: Dict[Optional[str], RankingParams] = {}
        for r in rk:
            by_type.setdefault(r.type, r)                                                       # groupBy(_.`type`).map(_._2.head)
        return list(by_type.values())

    def train(self, data: PreparedData):
        """URAlgorithm.train :292-307 (the URModel is returned instead of being written to Elasticsearch)."""
        if self.recsModel == "all":
            return self.calcAll(data)
        if self.recsModel == "collabFiltering":
            return self.calcAll(data, calcPopular=False)                                         # :296
        if self.recsModel == "backfill":
            raise NotImplementedError("recsModel=backfill re-ranks an EXISTING Elasticsearch index (calcPop :371-391 reads it back): needs ES, out of scope")
        raise ValueError(f"Bad algorithm param recsModel=[{self.recsModel}] in engine definition params, possibly a bad json value. "
                         "Use one of the available parameter values (all, collabFiltering, backfill).")  # :299-303

    def getRanks(self, fields: Dict[str, dict], now_ms: Optional[int] = None) -> Dict[str, Dict[str, float]]:
        """URAlgorithm.getRanksRDD :537-560 over the event store handed to the constructor."""
        from .pop_model import PopModel, getRanks
        if not self.eventStore:
            return {}
        if self.sess is None:
            import torch
            from .device import DeviceSession
            from . import _lib
            lib = self.library if self.library is not None else _lib.lib()
            self.sess = DeviceSession(torch.device("cuda", self.device) if torch.cuda.is_available() else torch.device("cpu"), lib)
        now_ms = int(time.time() * 1000) if now_ms is None else now_ms
        rankings = [{"name": r.name, "type": r.type, "eventNames": r.eventNames,
                     "duration_s": duration_seconds(r.duration if r.duration is not None else DefaultURAlgoParams.BackfillDuration),
                     "end_ms": _iso_ms(r.offsetDate)} for r in self.rankingsParams]
        return getRanks(rankings, PopModel(self.eventStore, fields, self.sess), self.modelEventNames, now_ms)

    def calcAll(self, data: PreparedData, calcPopular: bool = True, now_ms: Optional[int] = None):
        """URAlgorithm.calcAll :310-369: picks the call form, hands the matrices to the CCO build, zips the names back (:349),
        joins the item properties with the PopModel ranks (:351-358) and returns URModel(correlators, Seq(properties)) -- the
        object the reference calls .save on (:364-367)."""
        from .pop_model import propertiesWithRanks
        from .ur_model import URModel
        correlators = self.correlators(data)
        properties = propertiesWithRanks(data.fields, self.getRanks(data.fields, now_ms)) if calcPopular else {}   # :351-361
        return URModel(correlators, [properties])

    def correlators(self, data: PreparedData) -> List[Tuple[str, IndexedDataset]]:
        """The CCO half of calcAll (:321-349)."""
        ap = self.ap
        seed = ap.seed if ap.seed is not None else int(time.time() * 1000)                      # :240,:325 (default = wall clock)
        ids = [d for _, d in data.actions]
        if not ap.indicators:                                                                   # :322
            res = SimilarityAnalysis.cooccurrencesIDSs(
                ids, randomSeed=seed,
                maxInterestingItemsPerThing=_get_or_else(ap.maxCorrelatorsPerEventType, DefaultURAlgoParams.MaxCorrelatorsPerEventType),
                maxNumInteractions=_get_or_else(ap.maxEventsPerEventType, DefaultURAlgoParams.MaxEventsPerEventType),
                device=self.device, library=self.library, numGPUs=self.numGPUs)
        else:
            if len(ap.indicators) < len(ids):
                raise IndexError("indicators(i) is matched to the event matrices by position (URAlgorithm.scala:334-340)")
            datasets = [SimilarityAnalysis.DownsamplableCrossOccurrenceDataset(
                iD,
                _get_or_else(ap.indicators[i].maxItemsPerUser, DefaultURAlgoParams.MaxEventsPerEventType),
                _get_or_else(ap.indicators[i].maxCorrelatorsPerItem, DefaultURAlgoParams.MaxCorrelatorsPerEventType),
                ap.indicators[i].minLLR) for i, iD in enumerate(ids)]
            res = SimilarityAnalysis.crossOccurrenceDownsampled(datasets, seed, device=self.device, library=self.library, numGPUs=self.numGPUs)
        return list(zip([n for n, _ in data.actions], res))                                    # :349


    def train_events_on_device(self, trainingData, sess) -> List[Tuple[str, IndexedDataset]]:
        """Preparator.prepare + URAlgorithm.calcAll without leaving the GPU in between: the event streams are hashed on
        the host, dictionaries and matrices are built in HBM (Preparator.prepare_on_device) and those device matrices go
        straight into the CCO build (device.cross_occurrence_device); only the indicator rows come back.  Same model as
        `train(Preparator().prepare(trainingData))`."""
        from .device import DatasetParams, cross_occurrence_device
        from .preparator import Preparator
        from .similarity_analysis import _seed_to_int
        if self.recsModel not in ("all", "collabFiltering"):
            return self.train(Preparator().prepare(trainingData)).coocurrenceMatrices
        ap = self.ap
        pd, dp = Preparator().prepare_on_device(trainingData, sess, keep_on_device=True)
        seed = ap.seed if ap.seed is not None else int(time.time() * 1000)
        if not ap.indicators:
            params = [DatasetParams(_get_or_else(ap.maxEventsPerEventType, DefaultURAlgoParams.MaxEventsPerEventType),
                                    _get_or_else(ap.maxCorrelatorsPerEventType, DefaultURAlgoParams.MaxCorrelatorsPerEventType), None) for _ in dp.events]
        else:
            if len(ap.indicators) < len(dp.events):
                raise IndexError("indicators(i) is matched to the event matrices by position (URAlgorithm.scala:334-340)")
            params = [DatasetParams(_get_or_else(ap.indicators[i].maxItemsPerUser, DefaultURAlgoParams.MaxEventsPerEventType),
                                    _get_or_else(ap.indicators[i].maxCorrelatorsPerItem, DefaultURAlgoParams.MaxCorrelatorsPerEventType),
                                    ap.indicators[i].minLLR) for i in range(len(dp.events))]
        res = cross_occurrence_device(sess, [ev.matrix for ev in dp.events], params, _seed_to_int(seed))
        sess.synchronize()
        primary = pd.actions[0][1]
        out: List[Tuple[str, IndexedDataset]] = []
        for (name, ids), ind in zip(pd.actions, res):
            rp, ci, llr = ind.to_host()
            out.append((name, primary.create(rp, ci, primary.columnIDs, ids.columnIDs, llr)))
        return out


def toStringMap(indexedDataset: IndexedDataset, actionName: str) -> Dict[str, Dict[str, List[str]]]:
    """IndexedDatasetConversions.toStringMapRDD (package.scala:82-110): itemID -> {actionName: [ids, score desc]}.
    Rows arrive (llr desc, col asc) from the library; the reference's stable sortBy(-score) keeps that order.
    Items whose row is empty have no DRM row and therefore no entry."""
    out: Dict[str, Dict[str, List[str]]] = {}
    rp, ci = indexedDataset.row_ptr, indexedDataset.col_idx
    for i in range(indexedDataset.nrow):
        s, e = rp[i], rp[i + 1]
        if e > s:
            out[indexedDataset.rowIDs.inverse(i)] = {actionName: [indexedDataset.columnIDs.inverse(int(j)) for j in ci[s:e]]}
    return out
