#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one *_counter_collection.csv per pass) into per-kernel averages.

usage: pmc_summary.py out.json pass1_counter_collection.csv [pass2_counter_collection.csv ...]
Kernel names are shortened to the function name incl. template arguments.  When both FETCH_SIZE and WRITE_SIZE are
present the HBM bytes per launch follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): units KB, FETCH_SIZE
doubled on gfx950 (wide coalesced reads are tallied at half their size; gather-heavy kernels are over-corrected by up to 2x).
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = name.replace("urcco::", "")
    depth, out = 0, []
    for ch in name:          # cut the argument list: the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip()


def collect(paths):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    dur = defaultdict(lambda: [0.0, 0])
    for p in paths:
        seen = set()
        with open(p, newline="") as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                a = acc[k][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
                key = (p, row["Dispatch_Id"])
                if key not in seen:
                    seen.add(key)
                    d = dur[k]
                    d[0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                    d[1] += 1
    return acc, dur


def calibration(out_path, dirs):
    """FETCH_SIZE of tools/gather_microbench runs with a KNOWN byte count (one directory per run: its log sits beside it as <dir>.log):
    what does the counter report per algorithmic byte / per touched 64-byte sector / per touched 128-byte line for (a) a wide coalesced
    stream -- the case MI355X_MICROARCH.md calibrates: x2 -- and (b) random 48-byte rows walked by one lane each word after word (rows48dep: every load waits for its predecessor, as the SpGEMM's expand loop does;
    rows48 issues the row's twelve loads back to back: every one of them then misses and is tallied on its own)?"""
    import glob
    import os
    out = {"what": "rocprofv3 --pmc FETCH_SIZE on tools/gather_microbench (known byte counts), per launch; FETCH_SIZE in KB as rocprofv3 reports it", "cases": {}}
    for d in dirs:
        csvs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        log = [json.loads(l) for l in open(d + ".log") if l.startswith("{")]
        if not csvs or not log:
            continue
        acc, _ = collect(csvs)
        name = os.path.basename(d).replace("cal_", "")
        kern = {"stream": "stream_kernel", "rows48": "rows_kernel", "rows128": "rows_kernel", "rows48dep": "rows_dep_kernel"}[name]
        fetch_kb = acc[kern]["FETCH_SIZE"][0] / acc[kern]["FETCH_SIZE"][1]
        alg = float(log[0]["algorithmic_bytes_per_launch"])
        e = {"algorithmic_bytes": alg, "FETCH_SIZE_KB": round(fetch_kb, 1), "reported_bytes_over_algorithmic": round(fetch_kb * 1024 / alg, 4)}
        if name in ("rows48", "rows48dep"):     # 48-byte rows at a 48-byte pitch: 4 of every 8 rows straddle a 64-byte sector, 2 of 8 a 128-byte line
            rows = alg / 48
            e.update({"bytes_in_64B_sectors": rows * 96, "bytes_in_128B_lines": rows * 160,
                      "reported_over_64B_sector_bytes": round(fetch_kb * 1024 / (rows * 96), 4), "reported_over_128B_line_bytes": round(fetch_kb * 1024 / (rows * 160), 4)})
        out["cases"][name] = e
    st, r48 = out["cases"].get("stream"), out["cases"].get("rows48dep")
    if st and r48:
        out["stream_factor"] = round(1.0 / st["reported_bytes_over_algorithmic"], 3)
        # one lane per 48-byte row, 12 word loads: a load that misses the L2 is one tally of 64 bytes -> misses per row
        out["row_walk_l2_misses_per_row_of_12_loads"] = round(r48["FETCH_SIZE_KB"] * 1024 / 64 / (r48["algorithmic_bytes"] / 48), 2)
    json.dump(out, open(out_path, "w"), indent=1)
    print(f"calibration -> {out_path}")


def main():
    argv = sys.argv[1:]
    if argv and argv[0] == "--calibration":
        return calibration(argv[1], argv[2:])
    source_id, cal = None, None
    while argv and argv[0].startswith("--"):
        if argv[0] == "--source-id":
            source_id = argv[1]
        elif argv[0] == "--use-calibration":
            try:
                cal = json.load(open(argv[1]))
            except OSError:
                cal = None
        argv = argv[2:]
    out_path, paths = argv[0], argv[1:]
    acc, dur = collect(paths)
    # FETCH_SIZE -> bytes: x2.  The counter tallies every L2 -> fabric read request at 64 bytes although a request fills a 128-byte line:
    # measured on a wide coalesced stream (calibration "stream": exactly 0.5 of the bytes read, the guide's figure) AND on scattered rows
    # (calibration "rows48" / "rows48dep": ~12 of the 12 word loads of a 48-byte row miss when every lane of a full chip holds a line of
    # its own -- 65K lines per XCD against a 4 MB L2 -- and the counter reports 14.5x the algorithmic bytes = 12 x 64 / 48 x 0.9: one 64-byte
    # tally per miss).  So 2 x FETCH_SIZE is L2-MISS traffic in 128-byte lines, Infinity-Cache hits included -- an upper bound of the HBM bytes.
    f = (cal or {}).get("stream_factor", 2.0)
    res = {}
    for k, cs in acc.items():
        e = {c: round(v[0] / v[1], 1) for c, v in cs.items()}
        e["launches_profiled"] = max(v[1] for v in cs.values())
        e["avg_ns_profiled"] = round(dur[k][0] / max(dur[k][1], 1), 1)
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["fetch_factor"] = f
            e["fetch_factor_kind"] = "every L2 miss fills a 128-byte line and is tallied at 64 (calibrated on a stream and on scattered rows)"
            e["hbm_bytes_per_launch"] = int((f * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024)
        res[k] = e
    res = dict(sorted(res.items(), key=lambda kv: -kv[1]["avg_ns_profiled"] * kv[1]["launches_profiled"]))
    json.dump({"passes": paths, "kernel_source_id": source_id, "calibration": cal,
               "note": "per-launch averages; FETCH_SIZE / WRITE_SIZE in KB; hbm_bytes_per_launch = (fetch_factor * FETCH_SIZE + WRITE_SIZE) * 1024, fetch_factor per kernel (see calibration)",
               "kernels": res}, open(out_path, "w"), indent=1)
    print(f"{len(res)} kernels -> {out_path}")


if __name__ == "__main__":
    main()
