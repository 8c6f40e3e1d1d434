#!/bin/bash
# Library from a COPY of the tree's csrc/ with one or more patches applied -> OUT.so (the tree itself is not touched), for same-box A/B
# runs of candidate changes kept as patches under profiles/ (tools/candidates_ab.sh, URCCO_LIB=...).
# usage: tools/build_patch_variant.sh OUT.so profiles/a.patch [profiles/b.patch ...]
set -e
OUT=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
mkdir -p "$TMP/universal-recommender_amd"
cp -r "$ROOT/universal-recommender_amd/csrc" "$TMP/universal-recommender_amd/csrc"
cp "$ROOT/bench.py" "$TMP/bench.py"
cp -r "$ROOT/include" "$TMP/include"
for p in "$@"; do (cd "$TMP" && patch -p1 -s < "$ROOT/$p"); done
mkdir -p "$(dirname "$ROOT/$OUT")"
(cd "$TMP/universal-recommender_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -pthread \
   $(ls cco_*.hip) ingest_kernels.hip urcco_api.hip urcco_context.hip urcco_hash.hip -ldl -o "$ROOT/$OUT")
ls -la "$ROOT/$OUT"
