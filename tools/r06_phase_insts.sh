#!/bin/bash
# round 6: dynamic instruction counts of the SpGEMM classes per PHASE on config 4, counts-aboard form -- needs a measurement build of the library whose DBG
# instantiations exist for the packed form too and know two more bits (4194304 = expand + insert only, 2097152 = DBG instantiation with nothing switched off; + 16 no ranking, + 8 no select, + 4 no top-k, + 6 no LLR either):
#   tools/build_rows_variant.sh variants/dbgpk.so <the sed expressions in profiles/r06_spgemm_phase_instruction_counts.json "build">
# usage: URCCO_LIB=$PWD/variants/dbgpk.so tools/r06_phase_insts.sh [TAG]
O=gpurun_out/${1:-r06_phase_insts}; mkdir -p $O
export TMPDIR=/tmp
BITS="${BITS:-2097152 2097168 2097160 2097156 2097158 4194304 1}"
for d in $BITS; do
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $GRAFT_REPO_ROOT/$O/pmc_$d -o p -- python $GRAFT_REPO_ROOT/tools/ablate.py --config4 1.0 $d > $GRAFT_REPO_ROOT/$O/pmc_$d.log 2>&1)
  python tools/pmc_summary.py $O/insts_debug_$d.json $O/pmc_$d/*counter_collection.csv > /dev/null 2>&1
  rm -rf $O/pmc_$d
done
python - $O $BITS <<'PY'
import json, sys, os
O = sys.argv[1]
bits = [int(x) for x in sys.argv[2:]]
tab = {}
for d in bits:
    p = os.path.join(O, f"insts_debug_{d}.json")
    if not os.path.exists(p):
        continue
    j = json.load(open(p))
    for k, v in j["kernels"].items():
        if "cco_rows" in k and v.get("SQ_INSTS_VALU", 0) > 1e5:
            tab.setdefault(k, {})[d] = (v.get("SQ_INSTS_VALU", 0) / 1e6, v.get("SQ_INSTS_SALU", 0) / 1e6, v.get("SQ_INSTS_LDS", 0) / 1e6, v.get("avg_ns_profiled", 0) / 1e3)
print("kernel | debug: VALU M, SALU M, LDS M per launch (average over the 5 event types), us")
for k, row in sorted(tab.items()):
    print(k[:70], " | ".join(f"{d}: {x[0]:.1f} {x[1]:.1f} {x[2]:.1f} {x[3]:.0f}us" for d, x in row.items()))
json.dump({k: {str(d): dict(zip(("VALU_M", "SALU_M", "LDS_M", "avg_us"), x)) for d, x in row.items()} for k, row in tab.items()}, open(os.path.join(O, "phase_insts.json"), "w"), indent=1)
PY
