#!/bin/bash
# round 5: the full-size parity tests with the exchange route / eight emulated ranks, then the per-kernel profile of the emulated 8-rank build
O=gpurun_out/r05_tests_full; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -m gpu -x -q -k "full_config4 or full_config5 or partitioned_column" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o ks -- python $GRAFT_REPO_ROOT/bench.py --emulate-ranks 8 --workload config4 --steps 3 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
rm -f $O/prof/*kernel_trace.csv
grep "urcco::" $O/prof/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-200 | head -60
