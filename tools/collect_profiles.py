"""Copy the outputs of tools/final_measure.sh (gpurun_out/final, gpurun_out/pmcx_sq) into profiles/ under the round's names.
usage: python tools/collect_profiles.py <tag>      e.g. r03_v1"""
import collections, csv, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
F, P = os.path.join(ROOT, "gpurun_out", "final"), os.path.join(ROOT, "profiles")
pairs = [("bench_default.json", f"{tag}_bench_default.json"), ("bench_under_rocprof.json", f"{tag}_bench_under_rocprof.json"),
         ("kt/kt_kernel_stats.csv", f"{tag}_kernel_stats.csv"), ("bench_b512.json", f"{tag}_b512_graph_bench.json"),
         ("bench_fused.json", f"{tag}_fused_bf16x3_bench.json"), ("bench_fused_bf16.json", f"{tag}_fused_bf16_bench.json"),
         ("bench_layerwise.json", f"{tag}_layerwise_bench.json"), ("bench_fused_f16x2_t32.json", f"{tag}_fused_f16x2_t32_bench.json"),
         ("pmc_summary.json", "r04_pmc_summary.json"), ("pmc_summary_cfg2.json", "r04_pmc_summary_cfg2_bf16.json")]
for src, dst in pairs:
    s = os.path.join(F, src)
    if os.path.exists(s) and os.path.getsize(s) > 0:
        shutil.copy(s, os.path.join(P, dst))
        print("copied", dst)
    else:
        print("MISSING", src)
# SQ counters of the chain kernels (tools/pmc_pass.sh sq ...): per kernel sums over dispatches / launches
sq = os.path.join(ROOT, "gpurun_out", "pmcx_sq", "sq_counter_collection.csv")
if os.path.exists(sq):
    tot, disp = collections.defaultdict(lambda: collections.defaultdict(float)), collections.defaultdict(set)
    for r in csv.DictReader(open(sq)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "chain" not in k:
            continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    out = {}
    for k, c in tot.items():
        n = max(len(disp[k]), 1)
        e = {name: v / n for name, v in c.items()}
        e["launches"] = n
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "SQ_BUSY_CYCLES" in e:
            # SQ_BUSY_CYCLES: per-SE busy cycles summed over the chip's 32 shader engines; MFMA busy cycles summed over 1024 SIMDs
            e["mfma_busy_fraction_of_simd_time"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["SQ_BUSY_CYCLES"] / 32.0 * 1024.0)
        out[k] = e
    out["_note"] = ("rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU "
                    "SQ_LDS_BANK_CONFLICT -- python3 tools/pmc_chain.py 2 2 (every fused chain kernel at M = 524 288, fused_f16x2); per-launch means")
    json.dump(out, open(os.path.join(P, f"{tag}_pmc_sq_counters_f16x2.json"), "w"), indent=1)
    print("wrote", f"{tag}_pmc_sq_counters_f16x2.json")
