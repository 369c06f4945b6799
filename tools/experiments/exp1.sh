#!/bin/bash
# experiment: 8 waves in one workgroup vs two 4-wave workgroups per CU: timings + phase traces (trace builds)
cd "$(dirname "$0")/../.."
for v in "" _w8; do
  f=pano-nerf_amd/libpanonerf_hip$v.so
  echo "== $f"
  PN_LIB=$f timeout -k 10 120 python3 tools/experiments/check_chain.py 2 2>&1 | grep "fused forward"
  PN_LIB=$f timeout -k 10 120 python3 tools/experiments/check_chain_bwd.py 2 2>&1 | grep "fused dgrad\|fused backward\|fused tangent"
done
for v in _tr _w8tr; do
  f=pano-nerf_amd/libpanonerf_hip$v.so
  echo "== trace $f"
  PN_LIB=$f timeout -k 10 120 python3 tools/experiments/trace_chain.py 2 2>&1 | grep -v "^block\|^M=\|^  "
done
