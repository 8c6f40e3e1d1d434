#!/bin/bash
# round 5, final records on one box: the whole GPU suite, the default bench line (with the CPU baseline), rocprofv3 kernel stats + PMC traffic + SQ counters of the same code,
# the emulated 8-rank builds, config 5 and the exchange route
O=gpurun_out/r05_final; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_final.log 2> $O/bench_final.err; echo "bench rc=$?"; python tools/bench_brief.py $O/bench_final.log
tools/r05_measure.sh r05_final prof pmc_hbm pmc_sq > $O/measure.log 2>&1; tail -3 $O/measure.log
RANKS=8 tools/r05_emulate.sh > $O/emulate.log 2>&1; grep "rank_total" $O/emulate.log | cut -c1-200
timeout 300 python bench.py --workload config5 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_config5.log 2>/dev/null; python tools/bench_brief.py $O/bench_config5.log
timeout 300 python bench.py --force-exchange --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_force_exchange.log 2>/dev/null; python tools/bench_brief.py $O/bench_force_exchange.log
timeout 120 python tools/ablate.py --config4 1.0 0,4,1 > $O/ablation_topk.log 2>&1; grep debug= $O/ablation_topk.log
