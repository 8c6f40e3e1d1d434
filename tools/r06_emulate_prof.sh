#!/bin/bash
# round 6: rocprofv3 kernel stats of the emulated 8-rank build (all ranks on one device): which kernels make up the per-rank fixed costs
O=$PWD/gpurun_out/r06_emulate_prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ks -- python $GRAFT_REPO_ROOT/bench.py --emulate-ranks ${RANKS:-8} --workload ${WL:-config4} --steps 3 > $O/emulate.json 2> $O/emulate.err; echo "rc=$?"
find $O/prof -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
python - $O/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if 'urcco' in r['Name']]
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("urcco kernels total %.2f ms" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:60]:
    print("%-70s calls %6s total %9.3f ms avg %8.1f us" % (r['Name'].split('(')[0][-70:], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3))
PY
