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


def calc_all_with_ranks_case(sess, lib):
    """URAlgorithm.calcAll(calcPopular = true) on the reference's rank data through the reference's own engine file
    (examples/rank/rank-engine.json: parsed unchanged, `rankings` included): URModel(correlators, Seq(properties)) -> one
    document per item carrying its indicator lists AND its ranking fields (URAlgorithm.scala:351-367, URModel.scala:57-75);
    the popularRank values order the items as data/rank-test-query-expected.txt does.  recsModel = collabFiltering drops the
    ranks (calcAll(calcPopular = false), :296)."""
    from universal_recommender_amd.data_source import DataSource, DataSourceParams
    from universal_recommender_amd.preparator import Preparator
    from universal_recommender_amd.ur_algorithm import URAlgorithm, URAlgorithmParams, duration_seconds
    d, events, now = _rank_fixture()
    engine = {"datasource": {"params": {"appName": "default-rank", "eventNames": d["eventNames"]}},
              "algorithms": [{"name": "ur", "params": {"appName": "default-rank", "indexName": "urindex", "typeName": "items", "recsModel": "all",
                                                       "eventNames": d["eventNames"], "rankings": d["rankings"], "numGPUs": 1}}]}
    lines = [f"{e[0]},{e[1]},{e[2]}" for e in d["events"]] + ["product-9,$set,color:green"]
    td = DataSource(DataSourceParams.from_engine_json(engine)).readTraining(lines)
    ap = URAlgorithmParams.from_engine_json(engine)
    assert [r.type for r in ap.rankings] == ["popular", "userDefined", "random"] and ap.numGPUs == 1 and duration_seconds("3650 days") == 3650 * 86400
    ap.seed = 3
    algo = URAlgorithm(ap, library=lib, eventStore=events, sess=sess)
    assert [r.name for r in algo.rankingsParams] == ["popularRank", "defaultRank", "uniqueRank"]
    pd = Preparator().prepare(td)
    model = algo.calcAll(pd, now_ms=now + 1)
    docs = {x["id"]: x for x in model.documents()}
    ranks = PO.pop_calc("popular", events, ["show", "like"], 3650 * 86400, now + 1)
    assert {i: x["popularRank"] for i, x in docs.items() if "popularRank" in x} == ranks
    order = d["popular_order_expected"]
    by_rank = sorted(docs, key=lambda i: -docs[i].get("popularRank", 0.0))
    assert [docs[i].get("popularRank", 0.0) for i in by_rank] == [ranks.get(i, 0.0) for i in order if i in docs]   # the golden's other items only exist through $set events
    assert all("uniqueRank" in docs[i] for i in ranks) and not any("defaultRank" in x for x in docs.values())   # userDefined is empty
    assert docs["product-9"]["color"] == ["green"] and "uniqueRank" in docs["product-9"]      # properties-only item, full outer join
    corr = {n: m for n, m in model.coocurrenceMatrices}
    assert set(corr) == {"show", "like"} and any("show" in x or "like" in x for x in docs.values())
    # the default ranking (no `rankings` key): one all-time popularity ranking on the primary event, field popRank (:250-256)
    engine["algorithms"][0]["params"].pop("rankings")
    ap2 = URAlgorithmParams.from_engine_json(engine)
    ap2.seed = 3
    algo2 = URAlgorithm(ap2, library=lib, eventStore=events, sess=sess)
    docs2 = {x["id"]: x for x in algo2.calcAll(pd, now_ms=now + 1).documents()}
    assert {i: x["popRank"] for i, x in docs2.items() if "popRank" in x} == PO.pop_calc("popular", events, ["show"], 3650 * 86400, now + 1)
    ap2.recsModel = "collabFiltering"
    docs3 = {x["id"]: x for x in URAlgorithm(ap2, library=lib, eventStore=events, sess=sess).train(pd).documents()}
    assert not any("popRank" in x for x in docs3.values()) and "product-9" not in docs3
    # Some(Seq()) event names = every event name (PopModel.scala:194)
    from universal_recommender_amd.pop_model import PopModel
    pm = PopModel(events, {}, sess)
    assert pm.calc("popular", [], 3650 * 86400, now + 1) == PO.pop_calc("popular", events, [], 3650 * 86400, now + 1) == \
        PO.pop_calc("popular", events, ["show", "like"], 3650 * 86400, now + 1)


def test_calc_all_with_ranks_on_the_simulator(sim_session, sim_lib):
    calc_all_with_ranks_case(sim_session, sim_lib)


@pytest.mark.gpu
def test_calc_all_with_ranks_on_gpu(gpu_session):
    from universal_recommender_amd import _lib
    calc_all_with_ranks_case(gpu_session, _lib.load(_lib.DEFAULT_PATH))


def test_duration_and_date_parsing_follow_scala_duration_and_joda():
    """`duration` goes through scala.concurrent.duration.Duration(...).toSeconds.toInt (URAlgorithm.scala:542-543) and
    `offsetDate` / `endDate` through Joda's ISODateTimeFormat.dateTimeParser (PopModel.scala:66-74): every unit Duration accepts, number
    and unit with or without a blank, truncation to seconds, Int wrap; extended and basic ISO forms, any number of fraction digits,
    offsets with and without a colon, a string without an offset read in the process's local zone, a bad date -> None (the reference warns and uses now)."""
    import time
    from universal_recommender_amd.ur_algorithm import _iso_ms, duration_seconds
    assert duration_seconds("3650 days") == 315360000 and duration_seconds("90days") == 7776000 and duration_seconds("2 h") == 7200
    assert duration_seconds("1500 ms") == 1 and duration_seconds("999 millis") == 0 and duration_seconds("2000000 micros") == 2
    assert duration_seconds("3000000000 nanos") == 3 and duration_seconds("1.5 minutes") == 90 and duration_seconds("45 sec") == 45
    assert duration_seconds("100000 days") == (8640000000 + 2**31) % 2**32 - 2**31          # Long.toInt wraps
    for bad in ("ten days", "5 fortnights", ""):
        try:
            duration_seconds(bad)
            assert False, bad
        except ValueError:
            pass
    t = 1456920000000                                                                       # 2016-03-02T12:00:00Z
    assert _iso_ms("2016-03-02T12:00:00Z") == t and _iso_ms("2016-03-02T12:00:00.000Z") == t and _iso_ms("20160302T120000Z") == t
    assert _iso_ms("2016-03-02T04:00:00.000-08:00") == t and _iso_ms("2016-03-02T13:00:00+0100") == t and _iso_ms("2016-03-02T13:00+01") == t
    assert _iso_ms("2016-03-02T12:00:00.123456789Z") == t + 123 and _iso_ms("2016-03-02T12:00:00,5Z") == t + 500
    local = _iso_ms("2016-03-02T12:00:00")                                                   # no offset: the default (local) zone, as Joda
    assert local == int(time.mktime((2016, 3, 2, 12, 0, 0, 0, 0, -1))) * 1000
    assert _iso_ms("2016-03-02") == int(time.mktime((2016, 3, 2, 0, 0, 0, 0, 0, -1))) * 1000
    assert _iso_ms("yesterday") is None and _iso_ms("2016-13-45T00:00:00Z") is None and _iso_ms(None) is None
