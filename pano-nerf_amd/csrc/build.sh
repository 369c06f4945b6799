#!/bin/bash
# Build libpanonerf_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=../libpanonerf_hip.so
FLAGS="$PN_EXTRA --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
for f in pn_gemm pn_render pn_mlp pn_chain; do
  if [ ! -f "$f.o" ] || [ "$f.hip" -nt "$f.o" ] || [ pn_common.h -nt "$f.o" ] || [ ../../include/panonerf_hip.h -nt "$f.o" ]; then
    /opt/rocm/bin/hipcc $FLAGS -c "$f.hip" -o "$f.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" pn_gemm.o pn_render.o pn_mlp.o pn_chain.o
echo "built $OUT"
