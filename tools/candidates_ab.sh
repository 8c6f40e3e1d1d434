#!/bin/bash
# One gpurun call that prices the candidate patches of profiles/ against the tree's library (same box): per-class serialised times
# (tools/ablate.py), the row scan on the HBM-resident matrix (tools/rowscan_bench.py), the whole step (bench.py) and, for every variant,
# the output digest of one config-4 build (a candidate whose digest differs from the tree's is wrong, whatever its speed).
# Build the variants first, on the builder:  tools/build_patch_variant.sh tools/_variants/rows.so profiles/r04_candidate_rows_kernel_prefetch.patch
#                                            tools/build_patch_variant.sh tools/_variants/flags.so profiles/r04_candidate_flags_kernel_issue_order.patch
#                                            tools/build_patch_variant.sh tools/_variants/both.so profiles/r04_candidate_rows_kernel_prefetch.patch profiles/r04_candidate_flags_kernel_issue_order.patch
# then:  gpurun --timeout 300 -- 'tools/candidates_ab.sh TAG tools/_variants/rows.so tools/_variants/flags.so tools/_variants/both.so'
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
for v in cur "$@"; do
  n=$(basename $v .so)
  if [ $v = cur ]; then unset URCCO_LIB; else export URCCO_LIB=$PWD/$v; fi
  echo "== $n"
  echo "digest $(timeout 60 python tools/soak_once.py --mode multi --builds 1 2>/dev/null | grep 'DIGEST warmup' | md5sum | cut -c1-16)"
  timeout 60 python tools/ablate.py --config4 1.0 0 2>&1 | grep debug=
  timeout 60 python tools/rowscan_bench.py 2>&1 | tail -3
  timeout 90 python bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json, sys
for line in sys.stdin:
    if line.startswith('{\"metric\"'):
        j = json.loads(line); k = j['kernels']
        print('step %.3f ms  flags %.3f  micro %.3f  wave %.3f  block_small %.3f  block %.3f' % (j['ms_per_step'], k['downsample_flags']['ms_per_step'], k['cco_rows_micro']['ms_per_step'], k['cco_rows_wave']['ms_per_step'], k['cco_rows_block_small']['ms_per_step'], k['cco_rows_block']['ms_per_step']))"
done 2>&1 | tee $O/candidates_ab.log
