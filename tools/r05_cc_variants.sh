#!/bin/bash
# round 5: the part-local histogram's variants on config 4's raw matrices (totals per matrix; the partition pass is the same in all)
O=gpurun_out/r05_cc_variants; mkdir -p $O
for v in "URCCO_PL_LANES=16" "URCCO_PL_LANES=8" "URCCO_PL_LANES=4" "URCCO_PL_LANES=16 URCCO_PL_DEBUG=1" "URCCO_PL_LANES=16 URCCO_PL_DEBUG=2" "URCCO_PL_LANES=16 URCCO_PL_DEBUG=3" "URCCO_PL_LANES=4 URCCO_PL_DEBUG=1"; do
  echo "== $v"; env $v timeout 200 python tools/colcount_bench.py 1.0 --only-new 2>&1 | grep "part-local" | cut -c1-600
done 2>&1 | tee $O/variants.log
