#!/bin/bash
# round 5: the CSR row scan under both down-sampling RNGs (URCCO_RNG_SPLITMIX53 / URCCO_RNG_MIX32): timings on the HBM-resident matrix and on
# config 4's five matrices, then SQ counters of the flags kernel for each
O=gpurun_out/${1:-r05_rowscan_rng}; mkdir -p $O
for r in "" "--rng32"; do
  python tools/rowscan_bench.py --hbm 1.0 0 $r 2>&1 | grep debug
  python tools/rowscan_bench.py --config4 1.0 0 $r 2>&1 | grep debug
done | tee $O/timings.log
for r in splitmix53 mix32; do
  arg=""; [ $r = mix32 ] && arg="--rng32"
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
    i=$((i+1))
    (cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc $set -d $GRAFT_REPO_ROOT/$O/pmc_${r}_$i -o p -- python $GRAFT_REPO_ROOT/tools/rowscan_bench.py --hbm 1.0 0 $arg > $GRAFT_REPO_ROOT/$O/pmc_${r}_$i.log 2>&1)
  done
  python tools/pmc_summary.py $O/sq_counters_$r.json $O/pmc_${r}_*/*counter_collection.csv > /dev/null 2>&1
  python - $O/sq_counters_$r.json $r <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
for k, v in j["kernels"].items():
    if "downsample_flags" in k:
        print(sys.argv[2], k[:60], json.dumps(v))
PY
  rm -rf $O/pmc_${r}_1 $O/pmc_${r}_2
done
