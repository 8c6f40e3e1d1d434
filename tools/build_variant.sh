#!/bin/bash
# Library with csrc/ of git revision REV (everything else from the working tree) -> OUT, for same-box A/B runs (tools/lib_ab.sh)
# usage: [EXTRA_FLAGS=-D...] tools/build_variant.sh REV OUT.so
set -e
REV=$1; OUT=$2
D=$(mktemp -d)
git archive $REV universal-recommender_amd/csrc include | tar -x -C $D
cd $D/universal-recommender_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -pthread $EXTRA_FLAGS $(ls cco_*.hip) ingest_kernels.hip urcco_api.hip urcco_context.hip urcco_hash.hip -ldl -o $OLDPWD/$OUT
cd $OLDPWD; rm -rf $D; ls -la $OUT
