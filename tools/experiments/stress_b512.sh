#!/bin/bash
# repeat the 512-ray bench and print the loss of every run (a NaN was seen once)
cd "$(dirname "$0")/../.."
for i in $(seq 1 ${1:-12}); do
  timeout -k 10 100 python3 bench.py --global-batch 512 --steps 20 --warmup 5 --no-cpu-baseline --no-inference > gpurun_out/bs.json 2>/dev/null
  python3 -c "
import json
d=json.loads(open('gpurun_out/bs.json').read().strip().splitlines()[-1]); print($i, round(d['value']), d['loss'], d['config']['launch'])"
done
