#!/bin/bash
# rocprofv3 --kernel-trace --stats of the single-stream timed steps; prints the top kernels.  usage: tools/r06_prof.sh TAG [bench args]
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout -k 10 ${PROF_TIMEOUT:-200} rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o ks -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --single-stream --timed-only "$@" > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
rm -f $O/prof/*kernel_trace.csv
f=$(ls $O/prof/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $O/kernel_stats.csv && head -${TOP:-45} $f | cut -c1-200
