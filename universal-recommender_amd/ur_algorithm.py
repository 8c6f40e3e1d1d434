"""Host-side mirror of the train side of URAlgorithm (reference src/main/scala/URAlgorithm.scala:142-171 params,
:195-247 defaults, :292-369 train/calcAll) and of IndexedDatasetConversions.toStringMapRDD (package.scala:82-110).

Only the CCO model build is in scope: calcAll returns the per-event indicator matrices (what the reference hands to
URModel.save); PopModel ranks, Elasticsearch and the query side are out of scope."""
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

    @staticmethod
    def from_engine_json(engine: dict, name: str = "ur") -> "URAlgorithmParams":
        algos = [a for a in engine["algorithms"] if a.get("name") == name]
        if not algos:
            raise ValueError(f"no algorithm named {name!r} in engine.json")
        p = algos[0]["params"]
        inds = p.get("indicators")
        return URAlgorithmParams(
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
    def __init__(self, ap: URAlgorithmParams, device: int = 0, library=None):
        self.ap = ap
        self.device = device
        self.library = library
        self.recsModel = ap.recsModel or DefaultURAlgoParams.RecsModel
        if not ap.eventNames and not ap.indicators:                                            # :224-226
            raise ValueError("Must have either \"eventNames\" or \"indicators\" in algorithm parameters.")
        self.modelEventNames = [i.name for i in ap.indicators] if ap.indicators else list(ap.eventNames)  # :230-234

    def train(self, data: PreparedData) -> List[Tuple[str, IndexedDataset]]:
        """URAlgorithm.train :292-307 (the model is returned instead of being written to Elasticsearch)."""
        if self.recsModel in ("all", "collabFiltering"):
            return self.calcAll(data)
        if self.recsModel == "backfill":
            raise NotImplementedError("recsModel=backfill is the popularity-only retrain (PopModel): out of scope of the CCO path")
        raise ValueError(f"Bad algorithm param recsModel=[{self.recsModel}] in engine definition params, possibly a bad json value. "
                         "Use one of the available parameter values (all, collabFiltering, backfill).")  # :299-303

    def calcAll(self, data: PreparedData) -> List[Tuple[str, IndexedDataset]]:
        """URAlgorithm.calcAll :310-349: picks the call form, hands the matrices to the CCO build, zips names back."""
        ap = self.ap
        seed = ap.seed if ap.seed is not None else int(time.time() * 1000)                      # :240,:325 (default = wall clock)
        ids = [d for _, d in data.actions]
        if not ap.indicators:                                                                   # :322
            res = SimilarityAnalysis.cooccurrencesIDSs(
                ids, randomSeed=seed,
                maxInterestingItemsPerThing=_get_or_else(ap.maxCorrelatorsPerEventType, DefaultURAlgoParams.MaxCorrelatorsPerEventType),
                maxNumInteractions=_get_or_else(ap.maxEventsPerEventType, DefaultURAlgoParams.MaxEventsPerEventType),
                device=self.device, library=self.library)
        else:
            if len(ap.indicators) < len(ids):
                raise IndexError("indicators(i) is matched to the event matrices by position (URAlgorithm.scala:334-340)")
            datasets = [SimilarityAnalysis.DownsamplableCrossOccurrenceDataset(
                iD,
                _get_or_else(ap.indicators[i].maxItemsPerUser, DefaultURAlgoParams.MaxEventsPerEventType),
                _get_or_else(ap.indicators[i].maxCorrelatorsPerItem, DefaultURAlgoParams.MaxCorrelatorsPerEventType),
                ap.indicators[i].minLLR) for i, iD in enumerate(ids)]
            res = SimilarityAnalysis.crossOccurrenceDownsampled(datasets, seed, device=self.device, library=self.library)
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
            return self.train(Preparator().prepare(trainingData))
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
