#!/bin/bash
# round 5: per-kernel durations of the emulated 8-rank build of config 4 (rocprofv3 kernel trace; the trace keeps the order, so the 8-rank builds -- 4 of them -- can be
# told from the one-rank reference builds that follow)
O=gpurun_out/r05_emulate_prof; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o kt -- python $GRAFT_REPO_ROOT/bench.py --emulate-ranks 8 --workload config4 --steps 3 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
python - $O/prof/*kernel_trace.csv <<'PY' | tee $O/emulated_w8_kernel_table.txt
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "urcco::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the one-rank reference builds come last: they start at the first launch after the last pack_rows_kernel
last_pack = max(i for i, r in enumerate(rows) if "pack_rows" in r["Kernel_Name"])
multi = rows[: last_pack + 1]
# close the multi-rank section at the end of its last build: everything up to the next column-count partition after the last pack belongs to it; approximate by the last compact_indicators before the next pl_partition
nxt = next((i for i in range(last_pack + 1, len(rows)) if "pl_partition" in rows[i]["Kernel_Name"]), len(rows))
multi = rows[:nxt]
tot = collections.defaultdict(lambda: [0, 0.0])
for r in multi:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("urcco::", "")
    tot[n][0] += 1
    tot[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
builds, ranks = 4, 8
print("kernel, calls, us per rank and build (sum over the event types), share")
s = sum(v[1] for v in tot.values())
for n, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("%-70s %6d %9.1f %5.1f%%" % (n[:70], c, us / builds / ranks, 100 * us / s))
print("total per rank and build: %.1f us" % (s / builds / ranks))
PY
rm -f $O/prof/*kernel_trace.csv
