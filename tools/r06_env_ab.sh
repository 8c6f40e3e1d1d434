#!/bin/bash
# A/B of environment settings of the library on ONE box, one digest line (tools/bench_brief.py) per run:
#   tools/r06_env_ab.sh TAG REPS "NAME=VAL ..." "NAME=VAL ..." ...   ("-" = no setting)
TAG=$1; REPS=$2; shift 2
O=gpurun_out/$TAG; mkdir -p $O
for i in $(seq 1 $REPS); do
  j=0
  for v in "$@"; do
    j=$((j+1))
    if [ "$v" = "-" ]; then e=""; else e="$v"; fi
    env $e timeout 600 python bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-extras ${BENCH_ARGS} > $O/ab_${j}_$i.log 2>&1
    echo "[$v] $i rc=$? $(python tools/bench_brief.py $O/ab_${j}_$i.log)"
  done
done
