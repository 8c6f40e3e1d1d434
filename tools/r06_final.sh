#!/bin/bash
# round 6, final records on one box: rocprofv3 kernel stats + PMC traffic + SQ counters of the final sources, then the default bench line (with the CPU
# baseline; its roofline.traffic reads the PMC file just collected from the same sources), the emulated 8-rank builds, config 5 and the exchange route.
# The GPU test-suite is its own call (tools: `pytest tests -m gpu`).
O=gpurun_out/r06_final; mkdir -p $O
tools/r06_measure.sh r06_final prof pmc_hbm pmc_sq > $O/measure.log 2>&1; tail -3 $O/measure.log
cp $O/hbm_traffic_pmc_config4.json profiles/r06_hbm_traffic_pmc_config4.json   # (on the box: what the bench below quotes; copied into the tree again from gpurun_out)
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_final.log 2> $O/bench_final.err; echo "bench rc=$?"; python tools/bench_brief.py $O/bench_final.log
RANKS=8 tools/r06_emulate.sh > $O/emulate.log 2>&1; grep "rank_total" $O/emulate.log | cut -c1-200
timeout 300 python bench.py --workload config5 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_config5.log 2>/dev/null; python tools/bench_brief.py $O/bench_config5.log
timeout 300 python bench.py --force-exchange --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_force_exchange.log 2>/dev/null; python tools/bench_brief.py $O/bench_force_exchange.log
