#!/usr/bin/env python3
"""Per-kernel digest of a rocprofv3 --kernel-trace --stats csv (library kernels only): tools/prof_brief.py ks_kernel_stats.csv [steps]"""
import csv
import sys

steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "urcco" not in n:
        continue
    short = n.split("(")[0].replace("void ", "").replace("urcco::", "")
    ms = float(r["TotalDurationNs"]) / 1e6 / steps
    tot += ms
    print(f"{short[:58]:58s} calls/step {int(r['Calls']) / steps:6.1f} avg_us {float(r['AverageNs']) / 1e3:9.1f} ms/step {ms:8.3f}")
print(f"{'total':58s} {'':36s} ms/step {tot:8.3f}")
