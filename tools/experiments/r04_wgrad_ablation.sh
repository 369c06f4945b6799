#!/bin/bash
# timing ablations of the weight-gradient tiles (variant builds -DPN_ABL_WG=...: 1 no products, 2 no staging, 4 no barriers; WRONG results,
# never shipped): per-kernel durations from rocprofv3 --kernel-trace over tools/pmc_chain.py (one evaluation, M = 524 288 + second order)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for tag in "" _abl1 _abl2 _abl3 _abl7; do
  lib=$R/pano-nerf_amd/libpanonerf_hip$tag.so
  PN_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ablwg$tag -o x -- python3 $R/tools/pmc_chain.py 2 3 > $R/gpurun_out/ablwg$tag.log 2>&1
  echo "== build '${tag:-default}'"
  python3 - "$R/gpurun_out/ablwg$tag" <<'PY'
import csv, sys, glob
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*kernel_stats.csv")[0])))
for r in rows:
    if "k_chain_wgrad" in r["Name"]:
        print(f"   {r['Name'][:58]:58s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
done
