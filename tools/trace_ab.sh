#!/bin/bash
# kernel timelines of the timed region under two environment settings (one box): tools/trace_ab.sh TAG "ENV" "ENV" ...
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
j=0
for v in "$@"; do
  j=$((j+1))
  if [ "$v" = "-" ]; then e=""; else e="$v"; fi
  (cd /tmp && env $e timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/t$j -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras --timed-only > $GRAFT_REPO_ROOT/$O/t$j.log 2>&1)
  echo "[$v] rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/t$j.log | head -1)"
  f=$(ls $O/t$j/*kernel_trace.csv | head -1); python - "$f" $O/t$j.tsv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w") as o:
    for r in rows:
        o.write("\t".join([r["Queue_Id"], r.get("Stream_Id", ""), r["Start_Timestamp"], r["End_Timestamp"], r["Kernel_Name"][:60]]) + "\n")
PY
  rm -rf $O/t$j
done
