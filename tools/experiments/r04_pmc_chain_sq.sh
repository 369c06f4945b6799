#!/bin/bash
# where a chain kernel's wave cycles go: SQ counters over tools/pmc_chain.py 2 2 (two passes)
cd $GRAFT_REPO_ROOT
tools/pmc_pass.sh sqa 2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
tools/pmc_pass.sh sqb 2 SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD
tools/pmc_pass.sh sqc 2 SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_WAVE32_LDS
for n in sqa sqb sqc; do
  f=$(ls gpurun_out/pmcx_$n/*counter_collection.csv 2>/dev/null | head -1)
  [ -z "$f" ] && { echo "$n: no output"; tail -3 gpurun_out/pmcx_$n.log; continue; }
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if k.startswith("k_chain_") and "pack" not in k and "wexp" not in k:
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k[:46], {c: f"{sum(v)/len(v):.4g}" for c, v in acc[k].items()})
PY
done
