#!/usr/bin/env python
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace run (CSV): how much of a training step is the GPU
waiting for the next launch?   python tools/kernel_gaps.py <dir with *_kernel_trace.csv> [first kernel name of a step]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "k_front_gather"
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith(marker)]
if len(starts) < 3:
    print("no steps found (marker %s), %d kernels" % (marker, len(rows)))
    sys.exit(0)
# the last complete step
a, b = starts[-2], starts[-1]
step = rows[a:b]
span = step[-1][1] - step[0][0]
busy = sum(e - s for s, e, _ in step)
gaps = [(step[i + 1][0] - step[i][1], step[i][2][:40], step[i + 1][2][:40]) for i in range(len(step) - 1)]
pos = [g for g in gaps if g[0] > 0]
print("step of %d kernels: span %.3f ms, sum of kernel durations %.3f ms, idle between kernels %.3f ms (%d gaps, mean %.2f us, max %.1f us)" % (
    len(step), span / 1e6, busy / 1e6, sum(g[0] for g in pos) / 1e6, len(pos), sum(g[0] for g in pos) / max(1, len(pos)) / 1e3, max(g[0] for g in gaps) / 1e3))
print("overlapping launches (negative gap):", sum(1 for g in gaps if g[0] < 0))
for g in sorted(gaps, reverse=True)[:8]:
    print("  %7.1f us  after %-40s before %s" % (g[0] / 1e3, g[1], g[2]))
