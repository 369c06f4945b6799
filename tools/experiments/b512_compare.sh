#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/b512; mkdir -p $O
for m in fused fused_f16x2; do
  timeout -k 10 120 python3 bench.py --mlp-mode $m --global-batch 512 --steps 20 --warmup 5 --no-cpu-baseline --no-inference > $O/graph_$m.json 2> $O/graph_$m.err
  timeout -k 10 120 python3 bench.py --mlp-mode $m --global-batch 512 --steps 20 --warmup 5 --no-cpu-baseline --no-inference --graph off --streams 1 > $O/eager_$m.json 2> $O/eager_$m.err
done
for f in $O/*.json; do python3 tools/show_bench.py $f | head -14; done
