#!/bin/bash
# Library from the WORKING TREE's csrc/ with extra compiler flags -> OUT, for same-box A/B runs (tools/lib_ab.sh, URCCO_LIB=...)
# usage: tools/build_wt_variant.sh OUT.so -DURCCO_XLX_LDS=640 ...
set -e
OUT=$1; shift
cd universal-recommender_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -pthread "$@" $(ls cco_*.hip) ingest_kernels.hip urcco_api.hip urcco_context.hip urcco_hash.hip -ldl -o $OLDPWD/$OUT
cd $OLDPWD; ls -la $OUT
