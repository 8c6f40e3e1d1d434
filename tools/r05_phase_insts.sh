#!/bin/bash
# round 5: dynamic instruction counts of the SpGEMM classes per PHASE on config 4 -- SQ_INSTS_VALU / _SALU / _LDS of the row kernels with phases
# switched off (debug 0 = all; 4 = no top-k; 516 = no top-k, no count gather; 518 = + no LLR; 1 = gather only): differences = what a phase issues
O=gpurun_out/${1:-r05_phase_insts}; mkdir -p $O
export TMPDIR=/tmp
for d in 0 4 516 518 1; do
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $GRAFT_REPO_ROOT/$O/pmc_$d -o p -- python $GRAFT_REPO_ROOT/tools/ablate.py --config4 1.0 $d > $GRAFT_REPO_ROOT/$O/pmc_$d.log 2>&1)
  python tools/pmc_summary.py $O/insts_debug_$d.json $O/pmc_$d/*counter_collection.csv > /dev/null 2>&1
  rm -rf $O/pmc_$d
done
python - $O <<'PY'
import json, sys, glob, os
O = sys.argv[1]
tab = {}
for d in (0, 4, 516, 518, 1):
    p = os.path.join(O, f"insts_debug_{d}.json")
    if not os.path.exists(p):
        continue
    j = json.load(open(p))
    for k, v in j["kernels"].items():
        if "cco_rows" in k:
            tab.setdefault(k, {})[d] = (v.get("SQ_INSTS_VALU", 0) / 1e6, v.get("SQ_INSTS_SALU", 0) / 1e6, v.get("SQ_INSTS_LDS", 0) / 1e6, v.get("avg_ns_profiled", 0) / 1e3)
print("kernel | debug: VALU M, SALU M, LDS M per launch (average over the 5 event types), us")
for k, row in tab.items():
    print(k[:64], " | ".join(f"{d}: {x[0]:.1f} {x[1]:.1f} {x[2]:.1f} {x[3]:.0f}us" for d, x in sorted(row.items())))
PY
