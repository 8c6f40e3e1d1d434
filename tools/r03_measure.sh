#!/bin/bash
# Round-3 measurement driver for one gpurun call.  usage: tools/r03_measure.sh TAG section [section ...]
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
W=${WORKLOAD:-config4}
for s in "$@"; do
  echo "=== $s ($(date +%T))"
  case $s in
    valu)       timeout 300 tools/_build/valu_microbench > $O/valu_microbench.json 2> $O/valu.err; tail -c 600 $O/valu_microbench.json ;;
    tests)      timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest.log 2>&1; tail -20 $O/pytest.log ;;
    tests_k)    timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 -k "$TESTS_K" > $O/pytest_k.log 2>&1; tail -14 $O/pytest_k.log ;;
    bench)      timeout 1200 python bench.py > $O/bench.log 2>&1; tail -c 12000 $O/bench.log ;;
    bench_lite) timeout 900 python bench.py --workload $W --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-extras > $O/bench_lite_$W.log 2>&1; tail -c 5000 $O/bench_lite_$W.log ;;
    bench_fx)   timeout 900 python bench.py --workload $W --steps 10 --warmup 3 --force-exchange --no-cpu-baseline --no-extras > $O/bench_fx.log 2>&1; tail -c 1500 $O/bench_fx.log ;;
    bench_fxg)  timeout 900 python bench.py --workload $W --steps 10 --warmup 3 --force-exchange --gathered-primary --no-cpu-baseline --no-extras > $O/bench_fxg.log 2>&1; tail -c 1500 $O/bench_fxg.log ;;
    bench_c3)   timeout 900 python bench.py --workload config3 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_c3.log 2>&1; tail -c 5000 $O/bench_c3.log ;;
    bench_c5)   timeout 900 python bench.py --workload config5 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_c5.log 2>&1; tail -c 5000 $O/bench_c5.log ;;
    prof)       (cd /tmp && timeout -k 10 ${PROF_TIMEOUT:-150} rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$W -o ks -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-extras --single-stream --timed-only > $GRAFT_REPO_ROOT/$O/prof_$W.log 2>&1); rm -f $O/prof_$W/*kernel_trace.csv; ls $O/prof_$W; head -30 $O/prof_$W/*kernel_stats.csv ;;
    pmc_hbm)    for c in FETCH_SIZE WRITE_SIZE; do (cd /tmp && timeout -k 10 ${PROF_TIMEOUT:-150} rocprofv3 --kernel-trace --output-format csv --pmc $c -d $GRAFT_REPO_ROOT/$O/pmc_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-extras --single-stream --timed-only > $GRAFT_REPO_ROOT/$O/pmc_$c.log 2>&1); done
                python tools/pmc_summary.py $O/hbm_traffic_pmc_$W.json $O/pmc_FETCH_SIZE/*counter_collection.csv $O/pmc_WRITE_SIZE/*counter_collection.csv; rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE ;;
    pmc_sq)     i=0; for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"; do i=$((i+1));
                  (cd /tmp && timeout -k 10 ${PROF_TIMEOUT:-150} rocprofv3 --kernel-trace --output-format csv --pmc $set -d $GRAFT_REPO_ROOT/$O/pmc_sq$i -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-extras --single-stream --timed-only > $GRAFT_REPO_ROOT/$O/pmc_sq$i.log 2>&1); done
                python tools/pmc_summary.py $O/sq_counters_pmc_$W.json $O/pmc_sq*/*counter_collection.csv; rm -rf $O/pmc_sq1 $O/pmc_sq2 $O/pmc_sq3 ;;
    ab)         # same-box A/B of library builds: AB_LIBS="a.so b.so" (paths relative to the repo); current build = cur
                for i in $(seq 1 ${AB_REPS:-2}); do for v in cur $AB_LIBS; do n=$(basename $v .so); if [ $v = cur ]; then unset URCCO_LIB; else export URCCO_LIB=$PWD/$v; fi
                  timeout 600 python bench.py --workload $W --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-extras > $O/ab_${W}_${n}_$i.log 2>&1
                  echo "$W $n $i rc=$? $(python tools/bench_brief.py $O/ab_${W}_${n}_$i.log)"; done; done; unset URCCO_LIB ;;
  esac
done
