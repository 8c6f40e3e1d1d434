"""BM25-free restatement of the reference's query side, just enough to read its integration goldens.

The goldens (data/integration-test-expected.txt, data/integration-test-item-set-expected.txt) hold Elasticsearch
`_score`s, which pin the CCO model only through MEMBERSHIP: an item gets a positive score iff one of the query's
should-clauses hits one of its indicator fields (or a boosted property), and it appears at all only if it passes the
must / must_not clauses.  This module rebuilds those clauses from the reference's query builder:

  should   user history per event vs the same-named indicator field      URAlgorithm.scala:795-839 (getBiasedRecentUserActions)
           the query item's own indicator lists vs the same fields        URAlgorithm.scala:770-792 (getBiasedSimilarItems)
           itemSet items vs the primary event's field                     URAlgorithm.scala:645
           fields with bias > 0 (boost)                                   URAlgorithm.scala:844
  must     fields with bias < 0 (filter)                                  URAlgorithm.scala:854
  must_not fields with bias == 0                                          URAlgorithm.scala:863-864
           the user's primary-event items (blacklistEvents default)       URAlgorithm.scala:236, :743-752
           the query item (returnSelf = false) and the itemSet           URAlgorithm.scala:756, :760-761
           items outside [available, expires] / the query's dateRange    URAlgorithm.scala:888-953
Item dates follow examples/import_handmade.py:24-31,63-81 (day offsets relative to the import time).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Set

HANDMADE_ITEMS = ['Iphone 6', 'Ipad-retina', 'Nexus', 'Surface', 'Iphone 4', 'Galaxy', 'Iphone 5']  # import_handmade.py:63


def handmade_dates() -> Dict[str, Dict[str, float]]:
    """date / available / expires of each item in days relative to import time (import_handmade.py:24-31, :76-80)."""
    out = {}
    event_date = -2.4
    for item in HANDMADE_ITEMS:
        out[item] = {"date": event_date, "available": event_date - 2.0, "expires": event_date + 2.0}
        event_date += 0.8
    return out


def item_properties(sets: Sequence[Sequence[str]]) -> Dict[str, Dict[str, List[str]]]:
    """`item,$set,name:v1:v2` lines.  The importer stamps DEcreasing event times line by line (import_handmade.py:24,61),
    so the FIRST $set of a property in the file is the newest and wins in aggregateProperties."""
    props: Dict[str, Dict[str, List[str]]] = {}
    for item, payload in sets:
        parts = payload.split(":")
        props.setdefault(item, {}).setdefault(parts[0], parts[1:])
    return props


def positive_items(query: dict, model: Dict[str, Dict[str, List[str]]], all_items: Sequence[str], history: Dict[str, Dict[str, List[str]]],
                   props: Dict[str, Dict[str, List[str]]], primary_event: str, dates: Optional[Dict[str, Dict[str, float]]] = None):
    """(candidates passing must/must_not, subset with a should hit) for one query.
    model: item -> {event: [indicator ids]}; history: user -> {event: [item ids]} from the UNFILTERED event stream."""
    should_terms: Dict[str, Set[str]] = {}
    user = query.get("user")
    if user is not None and user in history:
        for ev, items in history[user].items():
            should_terms.setdefault(ev, set()).update(items)
    item = query.get("item")
    if item is not None and item in model:
        for ev, ids in model[item].items():
            should_terms.setdefault(ev, set()).update(ids)
    item_set = query.get("itemSet") or []
    if item_set:
        should_terms.setdefault(primary_event, set()).update(item_set)
    boosts, filters, excludes = [], [], []
    for f in query.get("fields", []) or []:
        (boosts if f["bias"] > 0 else filters if f["bias"] < 0 else excludes).append(f)
    must_not: Set[str] = set()
    if user is not None and user in history:
        must_not.update(history[user].get(primary_event, []))
    if item is not None:
        must_not.add(item)
    must_not.update(item_set)
    candidates, positives = [], []
    for it in all_items:
        if it in must_not:
            continue
        p = props.get(it, {})
        if any(set(p.get(f["name"], [])) & set(f["values"]) for f in excludes):
            continue
        if not all(set(p.get(f["name"], [])) & set(f["values"]) for f in filters):
            continue
        if dates is not None:
            d = dates.get(it)
            if d is not None and not (d["available"] <= 0.0 <= d["expires"]):
                continue
            dr = query.get("dateRange")
            if dr is not None and d is not None and not (-1.0 <= d[dr["name"]] <= 1.0):   # the script asks for [yesterday, tomorrow]
                continue
        candidates.append(it)
        doc = model.get(it, {})
        hit = any(set(doc.get(ev, [])) & terms for ev, terms in should_terms.items())
        hit = hit or any(set(p.get(f["name"], [])) & set(f["values"]) for f in boosts)
        if hit:
            positives.append(it)
    return candidates, positives
