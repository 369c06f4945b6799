#!/bin/bash
# Round 4, VERDICT r3 item 1: weight-gradient GEMMs beside the chains on disjoint CUs - same-box A/B of bench.py.
# usage (GPU box): bash tools/experiments/r04_overlap_ab.sh > gpurun_out/r04_overlap_ab.log
B="python bench.py --steps 10 --warmup 3 --no-inference --no-cfg2 --no-cpu-baseline"
show() { python -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], round(d['value']), 'rays/s', round(d['ms_per_step'],3), 'ms', d['config']['launch'])" "$1" "$2"; }
run() { name=$1; shift; $B "$@" > gpurun_out/r04_ab_$name.json 2> gpurun_out/r04_ab_$name.err || { echo "$name FAILED"; tail -5 gpurun_out/r04_ab_$name.err; return 1; }; show gpurun_out/r04_ab_$name.json "$name"; }
run off_a --overlap off &&
run on_112_144 --overlap on --chain-wgs 112 --wgrad-wgs 144 &&
run on_128_128 --overlap on --chain-wgs 128 --wgrad-wgs 128 &&
run on_96_160 --overlap on --chain-wgs 96 --wgrad-wgs 160 &&
run on_144_112 --overlap on --chain-wgs 144 --wgrad-wgs 112 &&
run on_160_96 --overlap on --chain-wgs 160 --wgrad-wgs 96 &&
run on_0_0 --overlap on --chain-wgs 0 --wgrad-wgs 0 &&
run off_b --overlap off &&
run b512_off --global-batch 512 --overlap off &&
run b512_on_112_144 --global-batch 512 --overlap on --chain-wgs 112 --wgrad-wgs 144 &&
run b512_on_128_128 --global-batch 512 --overlap on --chain-wgs 128 --wgrad-wgs 128 &&
run b512_on_0_0 --global-batch 512 --overlap on --chain-wgs 0 --wgrad-wgs 0
