#!/bin/bash
# round 5, GPU call 2: the part-local column counts against the bucket-contiguous form (same library, URCCO_COLCOUNT_GLOBAL_LAYOUT=1), after a parity pass
O=gpurun_out/r05_cc; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -x -q -k "logic_case or config3_scaled or tile_edges or config4_quarter or config5_tenth" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
STEPS=20 tools/env_ab.sh r05_cc 2 - URCCO_COLCOUNT_GLOBAL_LAYOUT=1
for f in $O/ab_*.log; do echo "$f: $(python tools/bench_brief.py $f)"; done
