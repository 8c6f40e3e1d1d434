#!/bin/bash
# A/B of builds of the library on ONE box (boxes differ by ~10 % in launch-bound regions): tools/lib_ab.sh TAG REPS other.so [other2.so ...]
TAG=$1; REPS=$2; shift 2
O=gpurun_out/$TAG; mkdir -p $O
for i in $(seq 1 $REPS); do
  for v in cur "$@"; do
    n=$(basename $v .so)
    if [ $v = cur ]; then unset URCCO_LIB; else export URCCO_LIB=$PWD/$v; fi
    timeout 600 python bench.py --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline --no-extras > $O/ab_${n}_$i.log 2>&1
    echo "$n $i rc=$? $(python - $O/ab_${n}_$i.log <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith('{"metric"'):
        j = json.loads(line)
        k = j["kernels"]
        rows = sum(v["ms_per_step"] for n, v in k.items() if n.startswith("cco_rows"))
        print(f'step {j["ms_per_step"]:.3f} unordered {j["unordered_rows"]["ms_per_step"]:.3f} rows {rows:.3f} tr {k["transpose"]["ms_per_step"]:.3f} cc {k["column_counts"]["ms_per_step"]:.3f} rw {k["row_work"]["ms_per_step"]:.3f} | ' + " ".join(f'{n.replace("cco_rows_", "")}={v["ms_per_step"]:.3f}' for n, v in k.items() if n.startswith("cco_rows")))
PY
)"
  done
done
