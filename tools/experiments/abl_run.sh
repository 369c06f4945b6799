#!/bin/bash
# times the forward chain of every ablation variant found (tools/build_variant.sh), same box, back to back
cd "$(dirname "$0")/../.."
for v in "" _abl1 _abl2 _abl3 _abl4 _abl7 _abl11 _abl16 _abl20; do
  f=pano-nerf_amd/libpanonerf_hip$v.so
  [ -f $f ] || continue
  echo "== $f"
  PN_LIB=$f timeout -k 10 120 python3 tools/experiments/check_chain.py 2 2>&1 | grep "fused forward"
  PN_LIB=$f timeout -k 10 120 python3 tools/experiments/check_chain_bwd.py 2 2>&1 | grep "fused dgrad\|fused backward"
done
