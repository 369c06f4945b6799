#!/bin/bash
cd "$(dirname "$0")/../.."
for v in "" _t32; do
  f=pano-nerf_amd/libpanonerf_hip$v.so
  echo "== $f"
  PN_LIB=$f timeout -k 10 120 python3 tools/experiments/check_chain.py 2 2>&1 | grep "fused forward"
  PN_LIB=$f timeout -k 10 120 python3 tools/experiments/check_chain_bwd.py 2 2>&1 | grep "fused dgrad\|fused backward\|fused tangent"
done
f=pano-nerf_amd/libpanonerf_hip_t32tr.so
echo "== trace $f"
PN_LIB=$f timeout -k 10 120 python3 tools/experiments/trace_chain.py 2 2>&1 | grep -v "^block\|^M=\|^  "
