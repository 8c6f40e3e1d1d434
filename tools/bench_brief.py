#!/usr/bin/env python3
"""One-line digest of a bench.py log (same-box A/B runs)."""
import json
import sys

for line in open(sys.argv[1]):
    if line.startswith('{"metric"'):
        j = json.loads(line)
        k = j["kernels"]
        g = lambda n: k.get(n, {}).get("ms_per_step", 0.0)
        rows = {n.replace("cco_rows_", ""): v["ms_per_step"] for n, v in k.items() if n.startswith("cco_rows")}
        un = j.get("unordered_rows") or {}
        print(f'step {j["ms_per_step"]:.3f} latency {j.get("single_build_latency_ms") or 0:.3f} unordered {un.get("ms_per_step", 0):.3f} | spgemm {sum(rows.values()):.3f} else {j["serialised_ms"]["everything_else"]:.3f} | '
              f'cc {g("column_counts"):.3f} flags {g("downsample_flags"):.3f} scan {g("downsample_scan"):.3f} compact {g("downsample_compact"):.3f} tr {g("transpose"):.3f} rw {g("row_work"):.3f} '
              f'bin {g("binning"):.3f} ci {g("compact_indicators"):.3f} | ' + " ".join(f"{n}={v:.3f}" for n, v in rows.items()))
        break
else:
    print("no bench line; tail:", open(sys.argv[1]).read()[-400:].replace("\n", " | "))
