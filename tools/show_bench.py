import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "rays/s %.0f  ms/step %.2f  loss %.5f  psnr %.2f  launch %s" % (d["value"], d["ms_per_step"], d["loss"], d["psnr_batch_db"], d["config"]["launch"]), d.get("inference"))
    r = d["roofline"]
    print("   dominant", r["kernel"], "bound %s achieved %.1f peak %.1f %s frac %.3f e2e %.3f" % (r.get("bound"), r["achieved"], r["peak"], r.get("unit"), r["frac"], r["end_to_end_frac"]),
          "" if not r.get("mfma") else " (mfma frac %.3f%s)" % (r["mfma"]["frac"], ", hbm frac %.3f" % r["hbm"]["frac"] if r.get("hbm") else ""))
    for k, v in r["other"].items():
        print("   %-16s total %8.2f ms  launches %5d  avg %8.1f us  %7.1f TF  frac %.3f" % (k, v["total_ms"], v["launches"], v["avg_launch_us"], v["tflops"], v["frac"]),
              "" if not v.get("hbm_gbs_from_pmc_traffic") else " HBM %.0f GB/s" % v["hbm_gbs_from_pmc_traffic"])
    if "cpu_baseline" in d:
        print("   cpu", d["cpu_baseline"])
