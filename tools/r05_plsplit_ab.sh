#!/bin/bash
# round 5: ids per average histogram block of the part-local column counts (URCCO_PL_BLOCK_IDS: 0 = S from the block target only, 49152 default): one rank and 8 emulated ranks
O=gpurun_out/r05_plsplit; mkdir -p $O
for v in 0 49152 98304; do
  export URCCO_PL_BLOCK_IDS=$v
  echo "== URCCO_PL_BLOCK_IDS=$v"
  python tools/colcount_bench.py 1.0 --only-new 2>&1 | grep part-local | cut -c1-120
  timeout 300 python bench.py --emulate-ranks 8 --workload config4 --steps 3 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rank max %.3f mean %.3f one-rank %.3f speedup %.3f | input phase max %.3f' % (j['max_rank_ms'], j['mean_rank_ms'], j['one_rank_serialised_ms'], j['implied_compute_only_speedup'], j['phases']['input (counts + row scan on the user shard)']['max_ms']))"
done 2>&1 | tee $O/plsplit.log
