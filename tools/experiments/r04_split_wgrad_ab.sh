#!/bin/bash
# Round 4: weight-gradient jobs as two concurrent halves (part 1 = six uniform trunk layers, part 2 = the rest; half the CUs each, two
# streams) against the sequential launches - same-box A/B at several batch sizes (graph replay, default mode).
B="python bench.py --steps 20 --warmup 5 --no-inference --no-cfg2 --no-cpu-baseline"
show() { python -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], round(d['value']), 'rays/s', round(d['ms_per_step'],3), 'ms', d['config']['launch'])" "$1" "$2"; }
run() { name=$1; shift; "$@" > gpurun_out/r04_sp_$name.json 2> gpurun_out/r04_sp_$name.err || { echo "$name FAILED"; tail -5 gpurun_out/r04_sp_$name.err; return 1; }; show gpurun_out/r04_sp_$name.json "$name"; }
python -m pytest tests/test_gpu_chain.py -x -q -k "wgrad or chain_backward" > gpurun_out/r04_sp_tests.txt 2>&1; tail -2 gpurun_out/r04_sp_tests.txt
run seq_512 $B --global-batch 512 --split-wgrad-rays 0 &&
run split_512 $B --global-batch 512 --split-wgrad-rays 4096 &&
run seq_512_b $B --global-batch 512 --split-wgrad-rays 0 &&
run split_512_b $B --global-batch 512 --split-wgrad-rays 4096 &&
run seq_1024 $B --global-batch 1024 --split-wgrad-rays 0 &&
run split_1024 $B --global-batch 1024 --split-wgrad-rays 4096 &&
run seq_2048 $B --global-batch 2048 --split-wgrad-rays 0 &&
run split_2048 $B --global-batch 2048 --split-wgrad-rays 4096 &&
run seq_4096 $B --split-wgrad-rays 0 &&
run split_4096 $B --split-wgrad-rays 4096
