#!/bin/bash
# weight-gradient GEMMs with and without the lean form of the 256 x 256 tile: time and error against the layer-wise fp32 path
cd "$(dirname "$0")/../.."
for rows in 0 65536; do
  echo "== PN_WGRAD_LEAN_ROWS=$rows"
  PN_WGRAD_LEAN_ROWS=$rows timeout -k 10 200 python3 tools/experiments/check_chain_bwd.py ${1:-2} 2>&1 | grep "fused wgrad\|grad \|^M="
done
