#!/bin/bash
# round 5: the select's finish ranks {above the cut bin} + {cut bin} ONCE (URCCO_MERGED_RANK) against ambiguous set + survivors ranked separately, one box
O=gpurun_out/r05_merged_rank_ab; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "logic_case or config3_scaled or config5_style or select_overlay or unordered or config2 or config1" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2
STEPS=20 tools/lib_ab.sh r05_merged_rank_ab 2 tools/_variants/nomerge.so
