#!/bin/bash
# round 5: the new default grids (8x for the four big SpGEMM classes, 4x for the 512/1024-thread classes, bounded by ~8 rows per team) against
# the 2x of rounds 2-4, one box: config 4, config 3, and the emulated 8-rank build (a rank has an eighth of the rows)
O=gpurun_out/r05_grid_factors_final; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "logic_case or config3_scaled or config5_style or unordered" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2
STEPS=20 tools/env_ab.sh r05_grid_factors_final 2 - URCCO_GRID_FACTORS=2,2,2,2,2,2
BENCH_ARGS="--workload config3" STEPS=20 tools/env_ab.sh r05_grid_factors_final_c3 2 - URCCO_GRID_FACTORS=2,2,2,2,2,2
for e in "" "URCCO_GRID_FACTORS=2,2,2,2,2,2"; do env $e timeout 300 python bench.py --emulate-ranks 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/emu_${e:-default}.log 2>&1; grep -o '"max_rank_ms": [0-9.]*, "mean_rank_ms": [0-9.]*' $O/emu_${e:-default}.log; grep -o '"one_rank_serialised_ms": [0-9.]*' $O/emu_${e:-default}.log; done
