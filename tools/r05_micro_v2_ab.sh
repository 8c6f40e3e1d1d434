#!/bin/bash
# round 5: second form of the micro class's row body (mark + prefix-max user search, owner-lane candidates, replicated ranking) and the
# wave-level table-only LLR on one box: tree, tree without the wave-level LLR in the accumulator classes, without it anywhere, and the
# first form of the micro class (URCCO_MICRO_V2=0)
O=gpurun_out/r05_micro_v2_ab; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "logic_case or config3_scaled or config5_style or select_overlay or unordered or config2 or config1" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2
STEPS=20 tools/lib_ab.sh r05_micro_v2_ab 2 tools/_variants/rows_nowavefast.so tools/_variants/all_nowavefast.so tools/_variants/micro_v1.so
