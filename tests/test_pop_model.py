"""PopModel (reference src/main/scala/PopModel.scala:55-179, URAlgorithm.scala:351-358,537-560): the
oracle restatement against the reference's rank golden, and the device interval histogram behind the host mirror against
the oracle (simulator here, MI355X with -m gpu)."""
import json
import os

import numpy as np
import pytest

from oracle import cco_oracle as PO

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DAY = 86_400_000


def _rank_fixture():
    d = json.load(open(os.path.join(GOLDEN, "rank.json")))
    now = 1_790_000_000_000
    return d, [(e[1], e[2], now + int(e[3] * DAY)) for e in d["events"]], now


def test_oracle_popular_order_matches_the_reference_golden():
    """data/rank-test-query-expected.txt, "popular item recs only": with no user and no item the order is the popularRank
    order (ties fall to the other ranking fields, which this golden does not isolate): counts must not increase along
    it, and the trailing items are the ones without any show / like event."""
    d, events, now = _rank_fixture()
    r = [x for x in d["rankings"] if x["type"] == "popular"][0]
    ranks = PO.pop_calc("popular", events, r["eventNames"], 3650 * 86400, now + 1)
    order = d["popular_order_expected"]
    counts = [ranks.get(i, 0.0) for i in order]
    assert counts == sorted(counts, reverse=True) and counts[0] == 6.0 and counts[1] == 5.0
    assert set(order[:6]) == set(ranks) and counts[6:] == [0.0, 0.0, 0.0]


def pop_model_case(sess):
    from universal_recommender_amd.pop_model import PopModel, getRanks, propertiesWithRanks
    # ---- the reference's own rank data
    d, events, now = _rank_fixture()
    pm = PopModel(events, {}, sess)
    for r in d["rankings"]:
        if r["type"] == "popular":
            got = pm.calc("popular", r["eventNames"], 3650 * 86400, now + 1)
            assert got == PO.pop_calc("popular", events, r["eventNames"], 3650 * 86400, now + 1)
    # ---- synthetic stream: every ranking type, interval edges (start inclusive / end exclusive), ids outside the names
    rng = np.random.default_rng(8)
    n = 60000
    items = [f"i{int(x)}" for x in np.minimum(rng.zipf(1.3, n), 3000)]
    names = rng.choice(["buy", "view", "like", "$set"], n, p=[0.3, 0.4, 0.25, 0.05])
    times = now - rng.integers(0, 30 * DAY, n)
    events = [(str(nm), None if nm == "$set" else it, int(t)) for nm, it, t in zip(names, items, times)]
    events += [("buy", "edge", now - 9 * DAY), ("buy", "edge", now), ("buy", "edge", now - 3 * DAY), ("buy", "edge", now - 6 * DAY)]
    fields = {"i1": {"color": ["red"]}, "only-props": {"color": ["blue"]}}
    pm = PopModel(events, fields, sess)
    for mtype in ("popular", "trending", "hot", "userDefined", "nonsense"):
        for names_, dur, end in ((["buy"], 9 * 86400, now), (["buy", "like"], 21 * 86400, now - 2 * DAY), (["view"], 7, now - 40 * DAY),
                                 (["like"], 3 * 86400 + 1, now)):
            assert pm.calc(mtype, names_, dur, end) == PO.pop_calc(mtype, events, names_, dur, end), (mtype, names_, dur)
    rankings = [{"name": "popularRank", "type": "popular", "eventNames": ["buy", "like"], "duration_s": 20 * 86400},
                {"type": "trending", "eventNames": ["view"], "duration_s": 10 * 86400, "end_ms": now - DAY},
                {"type": "hot", "duration_s": 12 * 86400}, {"name": "defaultRank", "type": "userDefined"}]
    got = getRanks(rankings, pm, ["buy", "view"], now)
    ref = PO.get_ranks(rankings, events, ["buy", "view"], now)
    assert got == ref and any("trendRank" in v for v in got.values()) and any("hotRank" in v for v in got.values())
    props = propertiesWithRanks(fields, got)
    assert props == PO.properties_with_ranks(fields, ref) and props["only-props"] == {"color": ["blue"]} and "popularRank" in props["i1"]
    rnd = pm.calc("random", [], 30 * 86400, now + 1)
    assert set(rnd) == {e[1] for e in events if e[1] is not None and now + 1 - 30 * 86400 * 1000 <= e[2] < now + 1} | set(fields)


def test_pop_model_on_the_simulator(sim_session):
    pop_model_case(sim_session)


@pytest.mark.gpu
def test_pop_model_on_gpu(gpu_session):
    pop_model_case(gpu_session)
