#!/bin/bash
# Round-6 measurement driver for one gpurun call.  usage: tools/r06_measure.sh TAG section [section ...]
#   calib    FETCH_SIZE of known byte counts: a wide coalesced stream (the guide's x2 case) and random 48-byte rows walked by one lane
#            each (the SpGEMM's B'-row access) -> $O/fetch_size_calibration.json
#   pmc_hbm  FETCH_SIZE / WRITE_SIZE per kernel of the FINAL library on the bench workload (two separate passes), stamped with the id of
#            the kernel sources (bench.py refuses a traffic file whose id is not the library's) and with the calibration
#   pmc_sq   SQ counters per kernel (four separate passes: issue, LDS, waits, memory instructions) -> $O/sq_counters_pmc_$W.json
#   prof     rocprofv3 --kernel-trace --stats of bench.py --single-stream --timed-only
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
W=${WORKLOAD:-config4}
SRCID=$(python -c "import bench; print(bench.kernel_source_id())")
for s in "$@"; do
  echo "=== $s ($(date +%T))"
  case $s in
    calib)   for t in stream rows48 rows48dep; do
               (cd /tmp && timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/$O/cal_$t -o p -- $GRAFT_REPO_ROOT/tools/_build/gather_microbench $t > $GRAFT_REPO_ROOT/$O/cal_$t.log 2>&1); done
             python tools/pmc_summary.py --calibration $O/fetch_size_calibration.json $O/cal_stream $O/cal_rows48 $O/cal_rows48dep; rm -rf $O/cal_stream $O/cal_rows48 $O/cal_rows48dep; cat $O/fetch_size_calibration.json ;;
    pmc_hbm) for c in FETCH_SIZE WRITE_SIZE; do (cd /tmp && timeout -k 10 ${PROF_TIMEOUT:-150} rocprofv3 --kernel-trace --output-format csv --pmc $c -d $GRAFT_REPO_ROOT/$O/pmc_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-extras --single-stream --timed-only > $GRAFT_REPO_ROOT/$O/pmc_$c.log 2>&1); done
             python tools/pmc_summary.py --source-id $SRCID --use-calibration profiles/r04_fetch_size_calibration.json $O/hbm_traffic_pmc_$W.json $O/pmc_FETCH_SIZE/*counter_collection.csv $O/pmc_WRITE_SIZE/*counter_collection.csv; rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE ;;
    pmc_sq)  i=0; for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD"; do i=$((i+1));
               (cd /tmp && timeout -k 10 ${PROF_TIMEOUT:-150} rocprofv3 --kernel-trace --output-format csv --pmc $set -d $GRAFT_REPO_ROOT/$O/pmc_sq$i -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-extras --single-stream --timed-only > $GRAFT_REPO_ROOT/$O/pmc_sq$i.log 2>&1); done
             python tools/pmc_summary.py --source-id $SRCID $O/sq_counters_pmc_$W.json $O/pmc_sq*/*counter_collection.csv; rm -rf $O/pmc_sq1 $O/pmc_sq2 $O/pmc_sq3 $O/pmc_sq4 ;;
    prof)    (cd /tmp && timeout -k 10 ${PROF_TIMEOUT:-150} rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$W -o ks -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-extras --single-stream --timed-only > $GRAFT_REPO_ROOT/$O/prof_$W.log 2>&1); rm -f $O/prof_$W/*kernel_trace.csv; ls $O/prof_$W; head -24 $O/prof_$W/*kernel_stats.csv ;;
  esac
done
