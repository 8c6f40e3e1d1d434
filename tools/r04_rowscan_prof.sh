#!/bin/bash
# round 4: the CSR row scan on the HBM-resident matrix -- cached flags kernel (0), plain (524288), and the cached kernel's ablations; then SQ counters of both
O=gpurun_out/${1:-r04_rowscan}; mkdir -p $O
python tools/rowscan_bench.py --hbm 1.0 0,524288,1048576,2097152,4194304,6291456,8388608,15728640 2>&1 | grep debug > $O/ablation.log
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc $set -d $GRAFT_REPO_ROOT/$O/pmc$i -o p -- python $GRAFT_REPO_ROOT/tools/rowscan_bench.py --hbm 1.0 0,524288 > $GRAFT_REPO_ROOT/$O/pmc$i.log 2>&1)
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
