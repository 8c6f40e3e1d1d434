#!/bin/bash
# round 6: env-knob sweep of the emulated 8-rank build (per-rank fixed costs): histogram block size, SpGEMM grid factors at shard scale
O=gpurun_out/r06_emulate_env; mkdir -p $O
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout 600 python bench.py --emulate-ranks 8 --workload config4 --steps 3 > $O/$tag.json 2> $O/$tag.err
  python - $O/$tag.json $tag <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s max %.3f mean %.3f | " % (sys.argv[2], j["max_rank_ms"], j["mean_rank_ms"]) + " ".join("%s %.3f" % (k.split(" ")[0][:8], v["mean_ms"]) for k, v in j["phases"].items()))
except Exception as e:
    print(sys.argv[2], "no JSON:", e)
PY
}
run base A=1







run base2 A=1
run cc_global URCCO_COLCOUNT_GLOBAL_LAYOUT=1
run pl_lanes16 URCCO_PL_LANES=16
run pl_lanes4 URCCO_PL_LANES=4
run chunk_small URCCO_PH_CHUNK_BIG_NNZ=1000000
run base3 A=1
