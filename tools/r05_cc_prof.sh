#!/bin/bash
# round 5: column counts alone (config 4's five raw matrices), both layouts, then per-kernel durations of the same under rocprofv3
O=gpurun_out/r05_cc_prof; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/colcount_bench.py 1.0 2>&1 | tee $O/colcount_bench.log
(cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o ks -- python $GRAFT_REPO_ROOT/tools/colcount_bench.py 1.0 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
rm -f $O/prof/*kernel_trace.csv
grep -h "urcco::p[hl]_\|scan_" $O/prof/*kernel_stats.csv | cut -c1-200
