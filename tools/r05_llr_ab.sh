#!/bin/bash
# round 5: single-check LLR + DBG instantiations against the previous commit's library, after a parity pass; then the emulated 8-rank build (new pack_rows)
O=gpurun_out/r05_llr_ab; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "logic_case or config3_scaled or config5_style or select_overlay or llr_and_rng" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2
STEPS=20 tools/lib_ab.sh r05_llr_ab 2 tools/_variants/head.so
for f in $O/ab_*.log; do echo "$f: $(python tools/bench_brief.py $f)"; done
RANKS=8 tools/r05_emulate.sh
