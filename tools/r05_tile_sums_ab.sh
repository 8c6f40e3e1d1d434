#!/bin/bash
# round 5: the fused expand pass leaves the scan-tile sums of the lengths (the expand scans skip their reduce pass) against the previous commit, one box
O=gpurun_out/r05_tile_sums_ab; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config3_scaled or config5_style or host_level or stream_per_event or back_to_back or fused_expand or unordered or exchange_path" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2
STEPS=20 tools/lib_ab.sh r05_tile_sums_ab 3 tools/_variants/head.so
for i in 1 2 3; do for v in cur head; do python tools/bench_brief.py $O/ab_${v}_$i.log | cut -c1-170; done; done
