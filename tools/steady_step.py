"""One steady-state iteration out of a rocprofv3 --kernel-trace: per-kernel calls and summed duration of the dispatches between
two occurrences of a marker kernel.  The --stats CSV next to it averages over every dispatch of the process, first launches
included (a handful of 20-30 ms outliers in the first iteration moved some averages by 10x in rounds 4/5); this is the table to
read per-step times from.
    python tools/steady_step.py <*_kernel_trace.csv> <marker substring> <markers per step> <step index> > out.csv"""
import collections
import csv
import re
import sys

path, marker, per_step, which = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
if len(marks) < per_step * (which + 1) + 1:
    sys.exit("only %d occurrences of %r: no step %d" % (len(marks), marker, which))
lo, hi = marks[per_step * which], marks[per_step * (which + 1)]
agg = collections.OrderedDict()
for r in rows[lo:hi]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    a = agg.setdefault(name, [0, 0])
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
span = int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])
busy = sum(v[1] for v in agg.values())
w = csv.writer(sys.stdout)
w.writerow(["kernel", "calls", "total_us", "avg_us", "percent_of_busy"])
w.writerow(["# one iteration: %d dispatches, wall %.3f ms, kernels busy %.3f ms (marker %r, step %d of the trace)"
            % (hi - lo, span / 1e6, busy / 1e6, marker, which), "", "", "", ""])
for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    w.writerow([name, n, round(t / 1e3, 1), round(t / 1e3 / n, 2), round(100.0 * t / busy, 2)])
