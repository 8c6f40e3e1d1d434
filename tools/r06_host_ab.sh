#!/bin/bash
# host-level (PCIe-inclusive) call on config 4 under several environment settings, one box: tools/r06_host_ab.sh TAG "ENV=.. ENV=.." "-" ...
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
j=0
for v in "$@"; do
  j=$((j+1))
  if [ "$v" = "-" ]; then e=""; else e="$v"; fi
  env $e URCCO_TRACE_HOST=1 timeout 600 python tools/host_level_c4.py config4 4 > $O/host_$j.log 2>&1
  echo "[$v] $(tail -1 $O/host_$j.log | cut -c1-200)"
done
