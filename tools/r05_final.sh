#!/bin/bash
# round 5, final records on one box: the whole GPU suite, rocprofv3 kernel stats + PMC traffic + SQ counters, then the default bench line (with the CPU
# baseline; its roofline.traffic reads the PMC file just collected from the same sources), the emulated 8-rank builds, config 5, the exchange route
# and the top-k ablation
O=gpurun_out/r05_final; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -2
tools/r05_measure.sh r05_final prof pmc_hbm pmc_sq > $O/measure.log 2>&1; tail -3 $O/measure.log
cp $O/hbm_traffic_pmc_config4.json profiles/r05_hbm_traffic_pmc_config4.json   # (on the box: what the bench below quotes; copied into the tree again from gpurun_out)
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_final.log 2> $O/bench_final.err; echo "bench rc=$?"; python tools/bench_brief.py $O/bench_final.log
RANKS=8 tools/r05_emulate.sh > $O/emulate.log 2>&1; grep "rank_total" $O/emulate.log | cut -c1-200
timeout 300 python bench.py --workload config5 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_config5.log 2>/dev/null; python tools/bench_brief.py $O/bench_config5.log
timeout 300 python bench.py --force-exchange --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_force_exchange.log 2>/dev/null; python tools/bench_brief.py $O/bench_force_exchange.log
timeout 120 python tools/ablate.py --config4 1.0 0,4,1 > $O/ablation_topk.log 2>&1; grep debug= $O/ablation_topk.log
