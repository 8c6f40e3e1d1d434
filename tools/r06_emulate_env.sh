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
run pl16k URCCO_PL_BLOCK_IDS=16384
run pl24k URCCO_PL_BLOCK_IDS=24576
run pl32k URCCO_PL_BLOCK_IDS=32768
run pl96k URCCO_PL_BLOCK_IDS=98304
run grid4 URCCO_GRID_FACTORS=4,4,4,4,2,2
run grid16 URCCO_GRID_FACTORS=16,16,16,16,8,8
run grid2 URCCO_GRID_FACTORS=2,2,2,2,2,2
run base2 A=1
