#!/bin/bash
# chain kernels of the default library and of every variant build: timings only
cd "$(dirname "$0")/../.."
for f in pano-nerf_amd/libpanonerf_hip.so pano-nerf_amd/libpanonerf_hip_*.so; do
  [ -f $f ] || continue
  echo "== $f"
  PN_LIB=$f timeout -k 10 120 python3 tools/experiments/check_chain.py 2 2>&1 | grep "fused forward"
  PN_LIB=$f timeout -k 10 120 python3 tools/experiments/check_chain_bwd.py 2 2>&1 | grep "fused dgrad\|fused backward\|fused tangent"
done
