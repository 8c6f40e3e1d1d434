#!/bin/bash
# Working-tree library with cco_rows.hip edited by a sed expression -> OUT.so (objects of the other sources are cached under /tmp/urcco_obj): same-box A/Bs of
# the row kernels' compile-time constants.  usage: tools/build_rows_variant.sh OUT.so 's/URCCO_G_WAVE = 2/URCCO_G_WAVE = 4/' [more sed expressions]
set -e
OUT=$1; shift
SRC=universal-recommender_amd/csrc
OBJ=/tmp/urcco_obj; mkdir -p $OBJ
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -pthread"
RFL="$FL $ROWS_FLAGS"  # (ROWS_FLAGS: extra compiler flags for cco_rows.hip only)
for f in cco_counts cco_rowscan cco_transpose cco_expand cco_misc ingest_kernels urcco_api urcco_context urcco_hash; do
  if [ ! -f $OBJ/$f.o ] || [ $SRC/$f.hip -nt $OBJ/$f.o ]; then echo $f; fi
done | xargs -r -P 4 -I{} /opt/rocm/bin/hipcc $FL -c $SRC/{}.hip -o $OBJ/{}.o
D=$(mktemp -d); cp $SRC/*.h $D/; cp $SRC/cco_rows.hip $D/
for e in "$@"; do sed -i "$e" $D/cco_rows.hip; done
diff $SRC/cco_rows.hip $D/cco_rows.hip | head -20 || true
/opt/rocm/bin/hipcc $RFL -I$PWD/include -Rpass-analysis=kernel-resource-usage -c $D/cco_rows.hip -o $D/cco_rows.o 2> $OUT.resources.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread $OBJ/*.o $D/cco_rows.o -ldl -o $OUT
rm -rf $D; ls -la $OUT
