#!/bin/bash
# kernel trace of the 512-ray step (eager launches so that every kernel shows with its own name): per-kernel table + the step's launch list
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace512; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt -- python3 $R/bench.py --global-batch 512 --graph off --steps 6 --warmup 3 --no-cpu-baseline --no-inference --no-cfg2 > $O/bench.json 2> $O/err.txt
ls $O
python3 - "$O" <<'PY'
import csv, sys, glob, os
O = sys.argv[1]
f = [p for p in glob.glob(O + "/**/*kernel_trace.csv", recursive=True)][0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last full step: find the last k_adam-like kernel boundaries
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "k_adam" in n]
a, b = idx[-2] + 1, idx[-1] + 1
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
busy = 0
prev_end = t0
print(f"launches in the step: {len(step)}")
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev_end) / 1e3:6.1f} gap  {(e - s) / 1e3:8.1f} us  {r['Kernel_Name'][:90]}")
    prev_end = e
print(f"step span {(prev_end - t0) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us, gaps {(prev_end - t0 - busy) / 1e3:.1f} us")
PY
