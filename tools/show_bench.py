import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.4g %s | %.3f ms/step | graph=%s" % (d["value"], d["unit"], d["ms_per_step"], d["config"].get("hip_graph")))
for k, v in d.get("kernels", {}).items():
    print("  %-32s %8.2f us  %8.1f %-8s frac %.3f" % (k, v["us"], v["achieved"], v["unit"], v["frac"]))
for k, v in d.get("kernel_families", {}).items():
    print("  family %-34s calls %4d  total %9.1f us  avg %8.2f us" % (k, v["calls"], v["total_us"], v["avg_us"]))
print("roofline:", {k: v for k, v in d.get("roofline", {}).items() if k != "note"})
print("cpu:", d.get("cpu_baseline"))
print("gpu eager:", d.get("gpu_eager_baseline"), "speedup", d.get("speedup_vs_gpu_eager"))
print("components:", d.get("components"))
