#!/bin/bash
# round 5: top-k bookkeeping of the LDS accumulator classes -- AND / OR of the keys by DPP ladders instead of ds_bpermute butterflies, list
# positions claimed per WAVE (ballot + one atomic, none in one-wave teams) instead of per lane: tree against the previous commit on one box
O=gpurun_out/r05_topk_claims_ab; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "logic_case or config3_scaled or config5_style or select_overlay or unordered or config2 or config1" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2
STEPS=20 tools/lib_ab.sh r05_topk_claims_ab 2 tools/_variants/head.so
