#!/bin/bash
# round 4: the CSR row scan on the HBM-resident matrix -- timings with parts switched off (32 cheap hash, 64 no threshold gather, 128 no row lookup), then
# two SQ counter passes.  (profiles/r04_rowscan_sq_counters.json and r04_rowscan_cached_kernel_ablation.log were taken with this script while the LDS-cached
# form of the flags kernel existed -- profiles/r04_negative_results_lds_threshold_cache_and_xlx_tables.patch --, whose own switches are gone with it.)
O=gpurun_out/${1:-r04_rowscan}; mkdir -p $O
python tools/rowscan_bench.py --hbm 1.0 0,32,64,128,224 2>&1 | grep debug > $O/ablation.log
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc $set -d $GRAFT_REPO_ROOT/$O/pmc$i -o p -- python $GRAFT_REPO_ROOT/tools/rowscan_bench.py --hbm 1.0 0 > $GRAFT_REPO_ROOT/$O/pmc$i.log 2>&1)
done
python tools/pmc_summary.py $O/sq_counters.json $O/pmc*/*counter_collection.csv > /dev/null 2>&1
python - $O/sq_counters.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
for k, v in j["kernels"].items():
    if "downsample_flags" in k:
        print(k, json.dumps(v))
PY
rm -rf $O/pmc1 $O/pmc2
cat $O/ablation.log
