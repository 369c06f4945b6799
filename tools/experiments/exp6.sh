#!/bin/bash
# the training step (bench.py, 10 steps) on the default library and on every variant build, same box
cd "$(dirname "$0")/../.."
for f in pano-nerf_amd/libpanonerf_hip.so pano-nerf_amd/libpanonerf_hip_*.so; do
  [ -f $f ] || continue
  echo "== $f"
  PN_LIB=$f timeout -k 10 200 python3 tools/bench_with_lib.py --no-cfg2 --no-cpu-baseline --no-inference --steps 10 --warmup 3 > /tmp/b.json 2>/tmp/b.err || tail -3 /tmp/b.err
  python3 tools/show_bench.py /tmp/b.json 2>/dev/null | grep "rays/s\|2, 4, 4, 2\|k_chain_fwd\|k_chain_bwd"
done
