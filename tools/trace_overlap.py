#!/usr/bin/env python3
"""Reads a rocprofv3 kernel_trace.csv and reports, for the LAST `frames` launches of k_blend_grouped, how the
kernels of the two queues interleave: per queue the busy time, and the time during which kernels of both queues
run at once.  (Does the GPU overlap two frames that are in flight on two streams?)"""
import csv
import sys
from collections import defaultdict

path, frames = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = list(csv.DictReader(open(path)))
rows = [r for r in rows if r["Kernel_Name"].startswith(("void gs::", "gs::"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
blends = [i for i, r in enumerate(rows) if "k_blend_grouped" in r["Kernel_Name"]]
first = blends[-frames] if len(blends) >= frames else blends[0]
# back up to the preprocess kernel that starts that frame
rows = rows[max(first - 40, 0):]
t0 = int(rows[0]["Start_Timestamp"])
by_q = defaultdict(list)
for r in rows:
    by_q[r["Queue_Id"]].append((int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r["Kernel_Name"][10:40]))
print("queues:", {q: len(v) for q, v in by_q.items()})
span = max(int(r["End_Timestamp"]) for r in rows) - t0
ev = []
for q, v in by_q.items():
    for s, e, _ in v:
        ev.append((s, 1))
        ev.append((e, -1))
ev.sort()
depth, last, t_by_depth = 0, 0, defaultdict(int)
for t, d in ev:
    t_by_depth[depth] += t - last
    last, depth = t, depth + d
print("span %.1f us; time with 0 / 1 / 2+ kernels running: %.1f / %.1f / %.1f us" % (
    span / 1e3, t_by_depth[0] / 1e3, t_by_depth[1] / 1e3, sum(v for k, v in t_by_depth.items() if k >= 2) / 1e3))
for r in rows[-60:]:
    print("%-4s %9.1f %9.1f  %s" % (r["Queue_Id"], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                   r["Kernel_Name"][10:60]))
