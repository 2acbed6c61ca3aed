#!/usr/bin/env python
"""Concurrency summary of a rocprofv3 --kernel-trace CSV: how much of the GPU-busy wall time had >= 2 kernels in
flight (side-stream weight gradients next to the main chain), and per-kernel mean durations.
usage: overlap_trace.py <dir with *kernel_trace.csv> [name filter for the 'side' kernels]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
side = sys.argv[2] if len(sys.argv) > 2 else "conv3x3_bwd_weight"
path = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
ev, dur, side_iv = [], defaultdict(list), []
for r in csv.DictReader(open(path)):
    s, e, n = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]
    ev.append((s, 1))
    ev.append((e, -1))
    short = n.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:60]
    dur[short].append(e - s)
    if side in n:
        side_iv.append((s, e))
ev.sort()
busy = multi = 0
depth, last = 0, ev[0][0]
for t, k in ev:
    if depth >= 1:
        busy += t - last
    if depth >= 2:
        multi += t - last
    depth += k
    last = t
print("trace: %s" % os.path.basename(path))
print("GPU busy %.2f ms, of which >= 2 kernels in flight %.2f ms (%.1f %%); sum of kernel durations %.2f ms"
      % (busy / 1e6, multi / 1e6, 100.0 * multi / max(busy, 1), sum(sum(v) for v in dur.values()) / 1e6))
print("side kernels (%s): %d, summed duration %.2f ms" % (side, len(side_iv), sum(e - s for s, e in side_iv) / 1e6))
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print("%-62s calls %5d  mean %9.1f us  total %8.2f ms" % (n, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
