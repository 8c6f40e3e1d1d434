#!/bin/bash
# round 5: scalar-register diet of the SpGEMM classes on one box: tree (own SGPR pairs for the hot arguments in every class + single-exit insert loop + fixed-trip search),
# with the old insert loop, with plain pointers in the micro class, both, and the previous commit
O=gpurun_out/r05_variants_ab2; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "logic_case or config3_scaled or config5_style or select_overlay" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2
STEPS=20 tools/lib_ab.sh r05_variants_ab2 2 tools/_variants/oldti.so tools/_variants/oldti_plainmicro.so tools/_variants/newti_plainmicro.so tools/_variants/head.so
