"""Per-dispatch view of one steady-state iteration of a rocprofv3 trace: every dispatch whose kernel name contains one of the
given substrings, in launch order, with its duration (kernel trace) or counter value (counter collection).  The 1x1 kernels of
the encoder's backward run last layer first, so the position in the list is the layer; the list is what shows WHICH channel
counts a family is slow at (wave imbalance over channel groups, rows that end inside a cache line, ...).
    python tools/per_layer_trace.py <*_kernel_trace.csv | *_counter_collection.csv> <marker> <step index> <substr> [<substr> ...]"""
import csv
import re
import sys

path, marker, which, subs = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4:]
rows = list(csv.DictReader(open(path)))
counters = "Counter_Name" in rows[0]
if counters:
    # one row per (dispatch, counter): fold into one row per dispatch
    by = {}
    for r in rows:
        d = by.setdefault(int(r["Dispatch_Id"]), {"Kernel_Name": r["Kernel_Name"], "id": int(r["Dispatch_Id"])})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    rows = [by[k] for k in sorted(by)]
else:
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
if len(marks) < which + 2:
    sys.exit("only %d occurrences of %r" % (len(marks), marker))
lo, hi = marks[which], marks[which + 1]
n = 0
for r in rows[lo:hi]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    if not any(s in name for s in subs):
        continue
    short = re.sub(r"\(.*", "", name.replace("void ", ""))
    if counters:
        vals = " ".join("%s=%.0f" % (k, v) for k, v in sorted(r.items()) if k not in ("Kernel_Name", "id"))
        print("%4d %-60s %s" % (n, short, vals))
    else:
        print("%4d %-60s %9.1f us" % (n, short, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    n += 1
