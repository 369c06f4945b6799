#!/bin/bash
# usage: tools/build_variant.sh <suffix> <extra hipcc flags...>: pn_chain.hip with extra defines -> libpanonerf_hip_<suffix>.so
set -e
cd "$(dirname "$0")/../pano-nerf_amd/csrc"
sfx=$1; shift
/opt/rocm/bin/hipcc "$@" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -c pn_chain.hip -o /tmp/pn_chain_$sfx.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpanonerf_hip_$sfx.so pn_gemm.o pn_render.o pn_mlp.o /tmp/pn_chain_$sfx.o
echo built $sfx
