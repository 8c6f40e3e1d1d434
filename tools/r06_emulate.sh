#!/bin/bash
# round 6: per-rank critical path of an 8-rank build on one GPU (bench.py --emulate-ranks), configs 4 and 5
O=gpurun_out/r06_emulate; mkdir -p $O
for w in config4 config5; do
  timeout 600 python bench.py --emulate-ranks ${RANKS:-8} --workload $w --steps 3 > $O/emulate_w${RANKS:-8}_$w.json 2> $O/emulate_w${RANKS:-8}_$w.err; echo "$w rc=$?"
  python - $O/emulate_w${RANKS:-8}_$w.json <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k: j[k] for k in ("rank_total_ms", "max_over_mean", "one_rank_serialised_ms", "implied_compute_only_speedup", "ideal_speedup_if_balanced", "pairs_max_over_mean", "bytes_received_per_rank_MB")})
    for k, v in j["phases"].items():
        print("  %-90s max %.3f mean %.3f (x%.3f) one-rank %.3f" % (k, v["max_ms"], v["mean_ms"], v["max_over_mean"] or 0, v["one_rank_ms"]))
except Exception as e:
    print("no JSON:", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
done
