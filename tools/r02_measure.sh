#!/bin/bash
# Round-2 measurement driver for one gpurun call.  usage: tools/r02_measure.sh TAG section [section ...]
# sections: tests tests_fast ablate rowscan bench bench_ss prof pmc_hbm pmc_sq
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
for s in "$@"; do
  case $s in
    tests)      timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; tail -15 $O/pytest.log ;;
    tests_fast) timeout 900 python -m pytest tests -m gpu -x -q --ignore=tests/test_gpu_scale.py > $O/pytest_fast.log 2>&1; tail -5 $O/pytest_fast.log ;;
    tests_k)    timeout 900 python -m pytest tests -m gpu -x -q -k "$TESTS_K" > $O/pytest_k.log 2>&1; tail -8 $O/pytest_k.log ;;
    ablate)     timeout 600 python tools/ablate.py 1.0 ${ABLATE:-0,1024,2048,4096,8192,15360} > $O/ablate.log 2>&1; cat $O/ablate.log ;;
    rowscan)    timeout 600 python tools/rowscan_bench.py 1.0 ${ROWSCAN:-0,32,64,128,224,4096} > $O/rowscan.log 2>&1; cat $O/rowscan.log ;;
    bench)      timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1; tail -c 9000 $O/bench.log ;;
    bench_fx)   timeout 900 python bench.py --steps 20 --warmup 5 --force-exchange --no-cpu-baseline --no-extras > $O/bench_fx.log 2>&1; tail -c 1500 $O/bench_fx.log ;;
    bench_fx5)  for i in 1 2 3 4 5; do timeout 600 python bench.py --steps 40 --warmup 5 --force-exchange --no-cpu-baseline --no-extras > $O/bench_fx_$i.log 2>&1; echo "rc=$?"; tail -c 400 $O/bench_fx_$i.log; done ;;
    bench_c4)   timeout 1200 python bench.py --steps 5 --warmup 2 --workload config4 --no-cpu-baseline --no-extras > $O/bench_c4.log 2>&1; tail -c 6000 $O/bench_c4.log ;;
    bench_c5)   timeout 1200 python bench.py --steps 5 --warmup 2 --workload config5 --no-cpu-baseline --no-extras > $O/bench_c5.log 2>&1; tail -c 3000 $O/bench_c5.log ;;
    bench_ss)   timeout 900 python bench.py --steps 20 --warmup 5 --single-stream --no-cpu-baseline --no-extras > $O/bench_ss.log 2>&1; tail -c 3000 $O/bench_ss.log ;;
    prof)       (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o ks -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --single-stream > $GRAFT_REPO_ROOT/$O/prof.log 2>&1); rm -f $O/prof/*kernel_trace.csv; ls $O/prof ;;
    pmc_hbm)    for c in FETCH_SIZE WRITE_SIZE; do (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $GRAFT_REPO_ROOT/$O/pmc_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --single-stream > $GRAFT_REPO_ROOT/$O/pmc_$c.log 2>&1); done
                python tools/pmc_summary.py $O/pmc_hbm.json $O/pmc_FETCH_SIZE/*counter_collection.csv $O/pmc_WRITE_SIZE/*counter_collection.csv; rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE ;;
    pmc_sq)     i=0; for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"; do i=$((i+1));
                  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $set -d $GRAFT_REPO_ROOT/$O/pmc_sq$i -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --single-stream > $GRAFT_REPO_ROOT/$O/pmc_sq$i.log 2>&1); done
                python tools/pmc_summary.py $O/pmc_sq.json $O/pmc_sq*/*counter_collection.csv; rm -rf $O/pmc_sq1 $O/pmc_sq2 $O/pmc_sq3 ;;
  esac
done
