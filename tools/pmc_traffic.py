"""Actual HBM traffic per kernel of a workload: mean FETCH_SIZE / WRITE_SIZE per dispatch (two rocprofv3 --pmc passes) joined
with the mean dispatch duration of a --kernel-trace pass of the same command; bytes = 2 x FETCH_SIZE + WRITE_SIZE KiB (gfx950,
MI355X_MICROARCH.md).  A kernel far below ~5 TB/s of ACTUAL traffic is not bound by HBM; one at ~5 TB/s is, and its algorithmic
bytes against this column say how much of what it moves is line granularity.
    python tools/pmc_traffic.py <dir FETCH pass> <dir WRITE pass> <dir kernel-trace pass> [min total ms]"""
import collections
import csv
import glob
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n).replace("void ", "")
    return re.sub(r"\(.*", "", n)[:110]


def counters(d):
    out = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        by = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            by[r["Dispatch_Id"]] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = r["Kernel_Name"]
        for k, v in by.items():
            out[short(names[k])].append(v)
    return out


fe, wr = counters(sys.argv[1]), counters(sys.argv[2])
dur = collections.defaultdict(list)
for f in glob.glob(sys.argv[3] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
floor = float(sys.argv[4]) if len(sys.argv) > 4 else 0.3
rows = []
for k in dur:
    if k not in fe or k not in wr:
        continue
    us = sum(dur[k]) / len(dur[k])
    f = 2048.0 * sum(fe[k]) / len(fe[k])
    w = 1024.0 * sum(wr[k]) / len(wr[k])
    rows.append((sum(dur[k]) / 1e3, k, len(dur[k]), us, f, w))
rows.sort(reverse=True)
print("%9s %6s %10s %10s %10s %8s  kernel" % ("total ms", "calls", "avg us", "fetch MB", "write MB", "TB/s"))
for tot, k, n, us, f, w in rows:
    if tot < floor:
        continue
    print("%9.2f %6d %10.1f %10.1f %10.1f %8.2f  %s" % (tot, n, us, f / 1e6, w / 1e6, (f + w) / us / 1e6, k))
