#!/bin/bash
# round 5: the primary's expand preparation folded into the fused pass of the secondaries (tree) against URCCO_FOLD_PRIMARY=0, one box
O=gpurun_out/r05_fold_primary_ab; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config3_scaled or config5_style or host_level or stream_per_event or back_to_back or fused_expand or unordered" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2
STEPS=20 tools/env_ab.sh r05_fold_primary_ab 3 - URCCO_FOLD_PRIMARY=0
for i in 1 2 3; do for j in 1 2; do python tools/bench_brief.py $O/ab_${j}_$i.log; done; done
