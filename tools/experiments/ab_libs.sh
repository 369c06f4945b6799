#!/bin/bash
# same-box A/B of library builds: tools/experiments/ab_libs.sh "<bench flags>" libA.so libB.so ... (alternating, two rounds)
flags=$1; shift
show() { python -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], round(d['value']), 'rays/s', round(d['ms_per_step'],3), 'ms', d['config']['launch'])" "$1" "$2"; }
for round in 1 2; do
  for lib in "$@"; do
    n=$(basename $lib .so)_r$round
    PN_LIB=$lib python tools/bench_with_lib.py --steps 10 --warmup 3 --no-inference --no-cfg2 --no-cpu-baseline $flags > gpurun_out/ab_$n.json 2> gpurun_out/ab_$n.err || { echo "$n FAILED"; tail -3 gpurun_out/ab_$n.err; }
    show gpurun_out/ab_$n.json "$n [$flags]"
  done
done
