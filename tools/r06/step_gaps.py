#!/usr/bin/env python3
"""Where one step's time goes outside its kernels: from a rocprofv3 --kernel-trace CSV, the LAST step's launches in start order (a step = from one
adamw burst's end to the next), every idle gap > MIN us on the busiest stream's timeline (gap = start - max end so far, so overlapped side-stream
kernels do not count as idle), and the totals.
    python tools/r06/step_gaps.py <kernel_trace.csv> [min_gap_us]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
MIN = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    return n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]


ev = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# step boundaries: the first ce_fwd_bwd launch of each step is a convenient anchor (one per step)
ce = [i for i, e in enumerate(ev) if e[0].startswith("ce_fwd_bwd")]
if len(ce) < 3:
    sys.exit("need at least three steps in the trace")
a, b = ce[-2], ce[-1]
seg = ev[a:b]
span = (seg[-1][2] - seg[0][1]) / 1e3
busy_end = seg[0][2]
idle = 0.0
kern = 0.0
print(f"one step, CE launch to CE launch: {len(seg)} launches, {span / 1e3:.2f} ms")
for j in range(1, len(seg)):
    n, s, e = seg[j]
    g = (s - busy_end) / 1e3
    if g > 0:
        idle += g
    if g > MIN:
        print(f"  idle {g:8.1f} us before {n:60s} (after {seg[j - 1][0]})")
    busy_end = max(busy_end, e)
for n, s, e in seg:
    kern += (e - s) / 1e3
print(f"sum of kernel durations {kern / 1e3:.2f} ms; idle on the merged timeline {idle / 1e3:.3f} ms; overlapped (sum - span + idle) {(kern - span + idle) / 1e3:.2f} ms")
