#!/bin/bash
# usage: r04_pmc_lib.sh <lib.so> <tag>: L2-side counters + kernel times of the chain kernels / weight-gradient tiles of ONE library build
lib=$1; tag=$2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PN_LIB=$R/$lib rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $R/gpurun_out/pmcy_$tag -o x -- python3 $R/tools/pmc_chain.py 2 2 > $R/gpurun_out/pmcy_$tag.log 2>&1
python3 - "$R/gpurun_out/pmcy_$tag" <<'PY'
import csv, sys, collections, glob
d = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(d + "/*counter_collection.csv")[0])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    if "k_chain_wgrad" in k:
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(acc):
    print(k[:60], {c: f"{sum(v)/len(v):.4g}" for c, v in acc[k].items()}, f"avg {sum(dur[k])/len(dur[k]):.1f} us (under pmc)")
PY
