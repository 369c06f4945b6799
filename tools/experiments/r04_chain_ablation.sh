#!/bin/bash
# timing ablations of the chain kernels on the round-4 tree (variant builds -DPN_ABL_CHAIN=...: bit 0 no refill DMA, 1 no fragment reads,
# 2 no MFMAs, 3 no ring barrier, 4 no T stores; WRONG results, never shipped): kernel durations from rocprofv3 over tools/pmc_chain.py 2 3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for tag in "" _cabl16 _cabl4 _cabl1 _cabl8 _cabl7 _cabl23; do
  lib=$R/pano-nerf_amd/libpanonerf_hip$tag.so
  PN_LIB=$lib timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ablch$tag -o x -- python3 $R/tools/pmc_chain.py 2 3 > $R/gpurun_out/ablch$tag.log 2>&1
  python3 - "$R/gpurun_out/ablch$tag" "${tag:-default}" <<'PY'
import csv, sys, glob
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*kernel_stats.csv")[0])))
d = {r["Name"].split("(")[0].replace("void ", ""): float(r["AverageNs"]) / 1e3 for r in rows if "k_chain_" in r["Name"] and "wgrad" not in r["Name"] and "pack" not in r["Name"] and "wexp" not in r["Name"]}
print(f"{sys.argv[2]:10s} " + "  ".join(f"{k[8:]:12s} {v:7.1f}" for k, v in sorted(d.items())))
PY
done
