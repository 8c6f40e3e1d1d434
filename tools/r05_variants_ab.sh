#!/bin/bash
# round 5: the SpGEMM instruction-diet variants on one box: tree (DBG instantiations + single-check LLR in the micro class + borrow-chain ranking),
# without the borrow chain, with the single-check LLR in every class, DBG instantiations only, and the previous commit
O=gpurun_out/r05_variants_ab; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "logic_case or config3_scaled or config5_style or select_overlay" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2
STEPS=20 tools/lib_ab.sh r05_variants_ab 2 tools/_variants/noasm.so tools/_variants/asm_fastrows.so tools/_variants/dbgonly.so tools/_variants/head.so
