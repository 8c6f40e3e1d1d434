#!/bin/bash
# FETCH_SIZE (rocprofv3 --pmc, L2 -> fabric read requests) of the SpGEMM classes with phases switched off (tools/ablate.py, config 4):
#   debug 0 = everything, 512 = no count gather (cB = 100), 1 = B'-row gather only.  The difference 0 - 512 is what the one scattered
#   2-byte count gather per candidate costs in line fills.   usage: tools/r06_fetch_by_phase.sh TAG
TAG=$1
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
for f in 0 512 1; do
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/$O/pmc_$f -o p -- python $GRAFT_REPO_ROOT/tools/ablate.py --config4 1.0 $f > $GRAFT_REPO_ROOT/$O/pmc_$f.log 2>&1)
  python tools/pmc_summary.py $O/fetch_debug_$f.json $O/pmc_$f/*counter_collection.csv > /dev/null 2>&1
  rm -rf $O/pmc_$f
done
python - $O <<'PY'
import json, sys
o = sys.argv[1]
d = {f: json.load(open(f"{o}/fetch_debug_{f}.json"))["kernels"] for f in (0, 512, 1)}
names = [k for k in d[0] if k.startswith("cco_rows")]
print("kernel | FETCH_SIZE MB per launch (as reported, x2 = 128-byte line fills): all phases / no count gather / B'-row gather only | avg us all / no gather / rows only")
for k in names:
    kd = k.replace("false>", "true>") if k.endswith("false>") else k   # the ablation switches live in the DBG instantiations
    g = lambda f: d[f].get(kd, d[f].get(k, {}))
    print(f"{k:48s} {d[0][k].get('FETCH_SIZE', 0) / 1024:9.1f} {g(512).get('FETCH_SIZE', 0) / 1024:9.1f} {g(1).get('FETCH_SIZE', 0) / 1024:9.1f} | "
          f"{d[0][k]['avg_ns_profiled'] / 1e3:8.1f} {g(512).get('avg_ns_profiled', 0) / 1e3:8.1f} {g(1).get('avg_ns_profiled', 0) / 1e3:8.1f}")
PY
