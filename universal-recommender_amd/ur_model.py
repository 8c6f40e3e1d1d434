"""Host-side mirror of URModel.save up to (not including) the Elasticsearch write (reference
src/main/scala/URModel.scala:47-102): the per-event indicator matrices become one document per item,

    {"id": <itemID>, "<event>": [indicator ids, strongest first], ..., <item properties>}

(`toStringMapRDD` per matrix, package.scala:82-110, then `groupAll`'s recursive cogroup + map merge, URModel.scala:87-102,
then `propsMap + ("id" -> itemId)`, :70-75).  The reference hands these maps to EsClient.hotSwap; here they are
returned, or written as NDJSON in Elasticsearch bulk format so that they can be indexed as they are.  ES itself, index
aliases and type mappings are out of scope."""
from __future__ import annotations

import json
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

from .indexed_dataset import IndexedDataset
from .pop_model import RankingFieldName
from .ur_algorithm import _iso_ms, toStringMap


def extractJvalue(dateNames: Sequence[str], key: str, value: object) -> object:
    """URModel.extractJvalue (URModel.scala:126-140), applied to every property of every document before indexing (:67-74):
    a list maps element-wise; a STRING under one of `dateNames` becomes a date (`new DateTime(s).toDate`; here a timezone-aware
    datetime), a string under a ranking field name (RankingFieldName.toSeq) becomes a double (`s.toDouble`); numbers, booleans
    and every other string pass through."""
    if isinstance(value, (list, tuple)):
        return [extractJvalue(dateNames, key, v) for v in value]
    if isinstance(value, str):
        if key in dateNames:
            from datetime import datetime, timezone
            ms = _iso_ms(value)
            if ms is None:
                raise ValueError(f"property {key!r}: {value!r} is not an ISO-8601 date (new DateTime(s) throws in the reference)")
            return datetime.fromtimestamp(ms / 1000.0, tz=timezone.utc)
        if key in RankingFieldName.toSeq():
            return float(value)
        return value
    return value


def _json_default(o):
    from datetime import datetime
    if isinstance(o, datetime):   # what a java.util.Date becomes on the wire to Elasticsearch: ISO-8601 with milliseconds (here always in UTC)
        return o.strftime("%Y-%m-%dT%H:%M:%S.") + f"{o.microsecond // 1000:03d}Z"
    raise TypeError(f"not JSON serialisable: {o!r}")


class URModel:
    def __init__(self, coocurrenceMatrices: Sequence[Tuple[str, IndexedDataset]], propertiesMaps: Sequence[Dict[str, Dict[str, object]]] = ()):
        self.coocurrenceMatrices = list(coocurrenceMatrices)
        self.propertiesMaps = list(propertiesMaps)

    def documents(self, dateNames: Sequence[str] = ()) -> List[Dict[str, object]]:
        """One map per item that has at least one indicator or property (groupAll is a full outer cogroup), every value through
        extractJvalue (URModel.scala:67-74)."""
        merged: Dict[str, Dict[str, object]] = {}
        for action_name, dataset in self.coocurrenceMatrices:
            for item, m in toStringMap(dataset, action_name).items():
                merged.setdefault(item, {}).update(m)
        for props in self.propertiesMaps:
            for item, m in props.items():
                merged.setdefault(item, {}).update(m)
        return [{**{k: extractJvalue(dateNames, k, v) for k, v in fields.items()}, "id": item} for item, fields in merged.items()]

    def save(self, path: str, dateNames: Sequence[str] = (), esIndex: str = "urindex", esType: str = "items") -> int:
        """URModel.save(dateNames, esIndex, esType) up to the ES write: NDJSON in ES bulk format, an action line then the document,
        per item.  Returns the number of documents."""
        docs = self.documents(dateNames)
        with open(path, "w") as f:
            for d in docs:
                f.write(json.dumps({"index": {"_index": esIndex, "_type": esType, "_id": d["id"]}}) + "\n")
                f.write(json.dumps(d, default=_json_default) + "\n")
        return len(docs)
