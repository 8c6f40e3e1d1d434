"""Host-side mirror of DataSource.readTraining for file-backed event streams (reference
src/main/scala/DataSource.scala:65-102).  The reference reads PEventStore; here the same `user,event,item` text the
reference's importers post (examples/import_handmade.py:34-48) is read directly.  HBase/ES are out of scope."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional, Sequence, Tuple


@dataclass
class DataSourceParams:
    """DataSource.scala:36-42."""
    appName: str = ""
    eventNames: List[str] = field(default_factory=list)
    eventWindow: Optional[dict] = None
    minEventsPerUser: Optional[int] = None

    @staticmethod
    def from_engine_json(engine: dict) -> "DataSourceParams":
        p = engine["datasource"]["params"]
        return DataSourceParams(p.get("appName", ""), list(p["eventNames"]), p.get("eventWindow"), p.get("minEventsPerUser"))


@dataclass
class TrainingData:
    """DataSource.scala:111-114: actions = Seq[(eventName, pairs of (userID, itemID))]."""
    actions: List[Tuple[str, List[Tuple[str, str]]]]
    fields: Dict[str, Dict[str, object]] = field(default_factory=dict)
    minEventsPerUser: Optional[int] = 1


class DataSource:
    def __init__(self, dsp: DataSourceParams):
        self.dsp = dsp

    def readTraining(self, lines: Iterable[str], delimiter: str = ",") -> TrainingData:
        by_event: Dict[str, List[Tuple[str, str]]] = {n: [] for n in self.dsp.eventNames}
        fields: Dict[str, Dict[str, object]] = {}
        for line in lines:
            line = line.rstrip("\r\n")
            if not line:
                continue
            data = line.split(delimiter)
            if data[1] == "$set":
                props = data[2].split(":")
                name = props.pop(0)
                fields.setdefault(data[0], {}).setdefault(name, props)  # first $set in the file is the newest event
                continue
            if data[1] in by_event:
                if not data[0] or not data[2]:
                    raise ValueError("Empty user or item ID")  # DataSource.scala:82
                by_event[data[1]].append((data[0], data[2]))
        # DataSource.scala:89: event types with no events are dropped
        actions = [(n, by_event[n]) for n in self.dsp.eventNames if by_event[n]]
        return TrainingData(actions, fields, self.dsp.minEventsPerUser)
