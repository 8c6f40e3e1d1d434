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


def main():
    out_path, paths = sys.argv[1], sys.argv[2:]
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
    res = {}
    for k, cs in acc.items():
        e = {c: round(v[0] / v[1], 1) for c, v in cs.items()}
        e["launches_profiled"] = max(v[1] for v in cs.values())
        e["avg_ns_profiled"] = round(dur[k][0] / max(dur[k][1], 1), 1)
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_launch"] = int((2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024)
        res[k] = e
    res = dict(sorted(res.items(), key=lambda kv: -kv[1]["avg_ns_profiled"] * kv[1]["launches_profiled"]))
    json.dump({"passes": paths, "note": "per-launch averages; FETCH_SIZE / WRITE_SIZE in KB; hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 correction)",
               "kernels": res}, open(out_path, "w"), indent=1)
    print(f"{len(res)} kernels -> {out_path}")


if __name__ == "__main__":
    main()
