#!/bin/bash
# cache-side counters of the weight-gradient tiles (is the 48-byte-unit load pattern re-fetching lines from L2?)
cd $GRAFT_REPO_ROOT
rocprofv3 --list-avail 2>/dev/null | grep -E "TCP_TCC_READ_REQ|TCP_TOTAL_CACHE|TCC_HIT|TCC_MISS|TCC_REQ|TCP_PENDING|TCC_EA0_RDREQ|TCP_TA_TCP_STATE_READ|TCP_GATE_EN|TA_BUSY|TCC_BUSY|TCP_TCC_NC_READ|TCP_TCC_UC_READ|TCP_TCC_CC_READ|TCP_TCC_RW_READ" | cut -c1-140 | sort -u | head -40
tools/pmc_pass.sh l2a 2 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum
tools/pmc_pass.sh l2b 2 TCC_EA0_RDREQ_sum TCC_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
for n in l2a l2b; do
  f=$(ls gpurun_out/pmcx_$n/*counter_collection.csv | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    if "k_chain" in k:
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k[:70], {c: f"{sum(v)/len(v):.4g}" for c, v in acc[k].items()})
PY
done
