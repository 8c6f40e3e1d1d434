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
from .ur_algorithm import toStringMap


class URModel:
    def __init__(self, coocurrenceMatrices: Sequence[Tuple[str, IndexedDataset]], propertiesMaps: Sequence[Dict[str, Dict[str, object]]] = ()):
        self.coocurrenceMatrices = list(coocurrenceMatrices)
        self.propertiesMaps = list(propertiesMaps)

    def documents(self) -> List[Dict[str, object]]:
        """One map per item that has at least one indicator or property (groupAll is a full outer cogroup)."""
        merged: Dict[str, Dict[str, object]] = {}
        for action_name, dataset in self.coocurrenceMatrices:
            for item, m in toStringMap(dataset, action_name).items():
                merged.setdefault(item, {}).update(m)
        for props in self.propertiesMaps:
            for item, m in props.items():
                merged.setdefault(item, {}).update(m)
        return [{**fields, "id": item} for item, fields in merged.items()]

    def save(self, path: str, esIndex: str = "urindex", esType: str = "items") -> int:
        """NDJSON, ES bulk format: an action line then the document, per item.  Returns the number of documents."""
        docs = self.documents()
        with open(path, "w") as f:
            for d in docs:
                f.write(json.dumps({"index": {"_index": esIndex, "_type": esType, "_id": d["id"]}}) + "\n")
                f.write(json.dumps(d) + "\n")
        return len(docs)
