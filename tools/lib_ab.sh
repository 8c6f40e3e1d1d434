#!/bin/bash
# A/B of two builds of the library on ONE box (boxes differ by ~10 % in launch-bound regions): tools/lib_ab.sh TAG other.so [reps]
TAG=$1; OTHER=$2; REPS=${3:-3}
O=gpurun_out/$TAG; mkdir -p $O
for i in $(seq 1 $REPS); do
  for v in cur other; do
    if [ $v = other ]; then export URCCO_LIB=$PWD/$OTHER; else unset URCCO_LIB; fi
    timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras > $O/ab_${v}_$i.log 2>&1
    echo "$v $i rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/ab_${v}_$i.log | head -1)"
  done
done
