#!/bin/bash
cd "$(dirname "$0")/.."
for v in "" _tile32; do
  f=pano-nerf_amd/libpanonerf_hip$v.so
  [ -f $f ] || continue
  echo "== $f"
  PN_LIB=$f timeout -k 10 120 python3 tools/check_chain.py 2 2>&1 | grep "fused forward"
  PN_LIB=$f timeout -k 10 120 python3 tools/check_chain_bwd.py 2 2>&1 | grep "fused dgrad\|fused backward\|fused tangent"
done
