"""Generates the committed golden fixtures from the reference's own test data (run in the build container, where
/root/reference exists; the GPU box only sees the JSON this writes).

  handmade.json            events of data/sample-handmade-data.txt (+ $set item properties), the engine params of
                           examples/handmade-engine.json, the 28 queries of examples/multi-query-handmade.sh and the
                           results data/integration-test-expected.txt holds for them
  item_sets.json           same for data/sample-handmade-item-set-data.txt / handmade-engine-item-sets.json /
                           multi-query-handmade-item-sets.sh / integration-test-item-set-expected.txt
  movielens.json           data/sample_movielens_data.txt split into buy / rate events exactly as
                           examples/import_movielens_eventserver.py does (random.seed(3), two randint draws per line)

  downsample.json          data/sample-downsamplable-data.txt with the engine params of examples/handmade-engine-downsample.json
                           (minEventsPerUser = 6): the reference ships no expected output for it -- an input-only fixture that
                           exercises the Preparator's user filter on the reference's own data

  rank.json                data/sample-rank-data.txt with the event times examples/rank/import_rank.py assigns, the `rankings` of
                           examples/rank/rank-engine.json and the popularity order data/rank-test-query-expected.txt holds

Usage: python tests/golden/make_golden.py [/root/reference]
"""
import json
import os
import random
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def parse_events(path):
    events, sets = [], []
    for line in open(path):
        line = line.rstrip("\r\n")
        if not line:
            continue
        d = line.split(",")
        if d[1] == "$set":
            sets.append([d[0], d[2]])
        else:
            events.append([d[0], d[1], d[2]])
    return events, sets


def parse_queries(path):
    """(title, query dict) for every curl in a multi-query script; the echo before a curl is its title."""
    text = open(path).read()
    out = []
    title = None
    pos = 0
    pat = re.compile(r'echo "([^"\n]*)"\s*\n|curl -H "Content-Type: application/json" -d (["\'])\s*\n(.*?)\n\}\2 http', re.S)
    for m in pat.finditer(text):
        if m.group(1) is not None:
            if m.group(1).strip() and not m.group(1).startswith("="):
                title = m.group(1)
            continue
        body = m.group(3) + "\n}"
        if m.group(2) == '"':
            body = body.replace('\\"', '"')
        body = re.sub(r'("\s*)\n(\s*")', r'\1,\n\2', body)      # the reference's pagination queries lack a comma
        body = body.replace("$BEFORE", "__TOMORROW__").replace("$AFTER", "__YESTERDAY__")
        out.append({"title": title, "query": json.loads(body)})
    return out


def parse_expected(path):
    out = []
    title = None
    for line in open(path):
        line = line.strip()
        if line.startswith('{"itemScores"'):
            out.append({"title": title, "itemScores": json.loads(line)["itemScores"]})
        elif line and not line.startswith("="):
            title = line
    return out


def handmade(name, data, engine, queries, expected):
    events, sets = parse_events(os.path.join(REF, "data", data))
    eng = json.load(open(os.path.join(REF, "examples", engine)))
    q = parse_queries(os.path.join(REF, "examples", queries))
    e = parse_expected(os.path.join(REF, "data", expected))
    assert len(q) == len(e), (len(q), len(e))
    for a, b in zip(q, e):
        assert a["title"] == b["title"], (a["title"], b["title"])
    doc = {"source": {"data": "data/" + data, "engine": "examples/" + engine, "queries": "examples/" + queries, "expected": "data/" + expected},
           "datasource_params": eng["datasource"]["params"], "algorithm_params": eng["algorithms"][0]["params"],
           "events": events, "sets": sets,
           "queries": [{"title": a["title"], "query": a["query"], "itemScores": b["itemScores"]} for a, b in zip(q, e)]}
    json.dump(doc, open(os.path.join(HERE, name), "w"), indent=1)
    print(name, len(events), "events", len(q), "queries")


def inputs_only(name, data, engine):
    events, sets = parse_events(os.path.join(REF, "data", data))
    eng = json.load(open(os.path.join(REF, "examples", engine)))
    doc = {"source": {"data": "data/" + data, "engine": "examples/" + engine, "expected": None},
           "datasource_params": eng["datasource"]["params"], "algorithm_params": eng["algorithms"][0]["params"],
           "events": events, "sets": sets}
    json.dump(doc, open(os.path.join(HERE, name), "w"), indent=1)
    print(name, len(events), "events (no reference golden)")


def rank():
    """data/sample-rank-data.txt as examples/rank/import_rank.py posts it: line k gets event time now - 0.8 days * k
    (import_rank.py:22-26,71; `$set` lines advance the clock too), stored here as day offsets so that the fixture does not
    depend on the day it is used.  Golden: the "popular item recs only" query of data/rank-test-query-expected.txt -- with
    no user and no item the result order is the popRank order of examples/rank/rank-engine.json (popular over show + like,
    3650 days)."""
    import re as _re
    events = []
    k = 0
    for line in open(os.path.join(REF, "data", "sample-rank-data.txt")):
        line = line.rstrip("\r\n")
        if not line:
            continue
        d = line.split(",")
        if d[1] != "$set":
            events.append([d[0], d[1], d[2], -0.8 * k])
        k += 1
    engine = json.load(open(os.path.join(REF, "examples", "rank", "rank-engine.json")))
    algo = [a for a in engine["algorithms"] if a["name"] == "ur"][0]["params"]
    text = open(os.path.join(REF, "data", "rank-test-query-expected.txt")).read()
    m = _re.search(r"query with no item or user id, ordered by popularity\s*\n\s*\n(\{.*?\})\s*\n", text, _re.S)
    expected = [x["item"] for x in json.loads(m.group(1))["itemScores"]]
    doc = {"source": {"data": "data/sample-rank-data.txt", "importer": "examples/rank/import_rank.py", "engine": "examples/rank/rank-engine.json",
                      "expected": "data/rank-test-query-expected.txt (popular item recs only)"},
           "rankings": algo["rankings"], "eventNames": algo["eventNames"], "events": events, "popular_order_expected": expected}
    json.dump(doc, open(os.path.join(HERE, "rank.json"), "w"), indent=1)
    print("rank.json", len(events), "events; expected popular order", expected)


def movielens():
    random.seed(3)                                   # import_movielens_eventserver.py:10,14
    events = []
    for line in open(os.path.join(REF, "data", "sample_movielens_data.txt")):
        d = line.rstrip("\r\n").split("::")
        ev = "rate" if random.randint(0, 1) == 1 else "buy"   # :21
        random.randint(0, 1)                                   # :37 category draw consumes the stream
        events.append([d[0], ev, d[1]])
    doc = {"source": {"data": "data/sample_movielens_data.txt", "importer": "examples/import_movielens_eventserver.py"},
           "eventNames": ["buy", "rate"], "events": events}
    json.dump(doc, open(os.path.join(HERE, "movielens.json"), "w"))
    print("movielens.json", len(events), sum(e[1] == "buy" for e in events), "buy", sum(e[1] == "rate" for e in events), "rate")


if __name__ == "__main__":
    handmade("handmade.json", "sample-handmade-data.txt", "handmade-engine.json", "multi-query-handmade.sh", "integration-test-expected.txt")
    handmade("item_sets.json", "sample-handmade-item-set-data.txt", "handmade-engine-item-sets.json", "multi-query-handmade-item-sets.sh",
             "integration-test-item-set-expected.txt")
    inputs_only("downsample.json", "sample-downsamplable-data.txt", "handmade-engine-downsample.json")
    movielens()
    rank()
