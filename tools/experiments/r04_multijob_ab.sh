#!/bin/bash
# Round 4: weight-gradient jobs of one tile configuration in ONE launch (grid.y = job) against one launch per job - same-box A/B.
B="python bench.py --steps 10 --warmup 3 --no-inference --no-cfg2 --no-cpu-baseline"
show() { python -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], round(d['value']), 'rays/s', round(d['ms_per_step'],3), 'ms', d['config']['launch'])" "$1" "$2"; }
run() { name=$1; shift; "$@" > gpurun_out/r04_mj_$name.json 2> gpurun_out/r04_mj_$name.err || { echo "$name FAILED"; tail -5 gpurun_out/r04_mj_$name.err; return 1; }; show gpurun_out/r04_mj_$name.json "$name"; }
run multi_4096 $B &&
run single_4096 env PN_WGRAD_SINGLE_JOBS=1 $B &&
run multi_512 $B --global-batch 512 &&
run single_512 env PN_WGRAD_SINGLE_JOBS=1 $B --global-batch 512 &&
run multi_4096_b $B &&
run multi_512_b $B --global-batch 512 &&
run multi_1024 $B --global-batch 1024 &&
run multi_2048 $B --global-batch 2048
