#!/bin/bash
# times the weight-gradient GEMMs (and checks them against the layer-wise path) for the default library and every variant build
cd "$(dirname "$0")/../.."
for f in pano-nerf_amd/libpanonerf_hip.so pano-nerf_amd/libpanonerf_hip_*.so; do
  [ -f $f ] || continue
  echo "== $f"
  PN_LIB=$f timeout -k 10 200 python3 tools/experiments/check_chain_bwd.py ${1:-2} 2>&1 | grep "fused wgrad\|grad layers.3.w\|grad layers.5.w\|grad view.w\|grad layers.0.b\|^M="
done
