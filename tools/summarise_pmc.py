"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel: mean counter value per dispatch."""
import csv, glob, re, sys, collections
out = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for f in glob.glob(path + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(\w+_kernel)(<[^>]*>)?", r["Kernel_Name"])
            name = (m.group(1) + (m.group(2) or "")).replace(",", ";") if m else r["Kernel_Name"][:40].replace(",", ";")
            out[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted(out)
counters = sorted({c for n in names for c in out[n]})
print("kernel,dispatches," + ",".join("mean_" + c for c in counters))
for n in names:
    if not n.startswith(("conv", "bn_", "grad_", "sinkhorn", "sg_", "head_", "reduce", "permute", "pool_")):
        continue
    nd = max(len(v) for v in out[n].values())
    print(n + "," + str(nd) + "," + ",".join("%.1f" % (sum(out[n][c]) / len(out[n][c])) if out[n][c] else "" for c in counters))
