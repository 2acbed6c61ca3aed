"""Print the per-family table of a bench JSON line:  python tools/fam.py gpurun_out/x.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("img/s %.1f  ms/step %.2f" % (d["value"], d["ms_per_step"]))
for r in d.get("kernel_families", []):
    print("  %-34s %7.3f ms/step  %6.1f TF/s  %6.0f GB/s" % (r["kernel"][:34], r["ms_per_step"], r["tflops"] or 0, r["algorithmic_GBps"] or 0))
