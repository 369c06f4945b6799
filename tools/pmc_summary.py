"""Per-launch HBM traffic of the GEMM kernels from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; csv output).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out/f -o f -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-inference
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out/w -o w -- python3 bench.py ... (same)
    python tools/pmc_summary.py out/f/f_counter_collection.csv out/w/w_counter_collection.csv > profiles/r01_pmc_summary.json

FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 tallies the 128-B requests of a wide streaming read at 64 B);
WRITE_SIZE is taken as is (16-B-per-lane stores).  Units: the counters report KB.
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    tot, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        tot[k] += float(r["Counter_Value"])
        n[k] += 1
    return {k: (tot[k] / n[k], n[k]) for k in tot}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
taken_on = sys.argv[3] if len(sys.argv) > 3 else None
out = {}
for k in fetch:
    if "gemm" not in k and "chain" not in k:
        continue
    f, n = fetch[k]
    w = write.get(k, (0.0, 0))[0]
    out[k] = {"launches": n, "fetch_size_raw_bytes_per_launch": f * 1024, "fetch_bytes_per_launch_x2_corrected": 2 * f * 1024,
              "write_bytes_per_launch": w * 1024, "hbm_bytes_per_launch": (2 * f + w) * 1024}
if "k_gemm_nt_dma" in out:
    out["k_gemm_nt"] = out["k_gemm_nt_dma"]  # bench.py's name for the NT class
# bench.py's class names for the fused kernels: launch-weighted means over the template instantiations
for cls in ("k_chain_fwd", "k_chain_dgrad", "k_chain_tangent", "k_chain_bwd", "k_chain_wgrad"):
    inst = [v for k, v in out.items() if k.startswith(cls + "<")]
    if inst:
        n = sum(v["launches"] for v in inst)
        out[cls] = {key: sum(v[key] * v["launches"] for v in inst) / n for key in inst[0] if key != "launches"}
        out[cls]["launches"] = n
out["_note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes of `bench.py --steps 1 --warmup 1 "
                "--no-cpu-baseline --no-inference` (global batch 4096, N=128); averages over all launches of the kernel; "
                "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B; the doubled figure "
                "matches the algorithmic A-operand bytes of the LDS-DMA loads); WRITE_SIZE as is (16-B-per-lane stores).")
out["_taken_on"] = taken_on  # box / date / workload size of the two passes (tools/final_measure.sh)
if len(sys.argv) > 4:  # training steps the passes ran: HBM bytes of ONE step over every kernel (the small ones included)
    steps = float(sys.argv[4])
    allk = set(fetch) | set(write)
    out["_step_total_bytes"] = sum((2 * fetch.get(k, (0.0, 0))[0] * fetch.get(k, (0.0, 0))[1] + write.get(k, (0.0, 0))[0] * write.get(k, (0.0, 0))[1])
                                   for k in allk) * 1024 / steps
    out["_steps_in_the_passes"] = steps
print(json.dumps(out, indent=1))
