#!/usr/bin/env python3
"""Per-launch durations of the 256x256 GEMM inside the step, from a rocprofv3 --kernel-trace CSV: the persistent kernel always has 256
workgroups, so the launches of one instantiation are told apart by their position in the layer's launch sequence (duration clusters).
    python tools/r04/trace_shapes.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = int(rows[-1]["End_Timestamp"])
# keep the last ~40 % of the trace (steady-state steps)
t0 = int(rows[0]["Start_Timestamp"])
cut = t0 + 0.6 * (t_end - t0)
by = collections.defaultdict(list)
for r in rows:
    if int(r["Start_Timestamp"]) < cut:
        continue
    n = r["Kernel_Name"]
    if "gemm256_kernel" not in n and "attn_" not in n:
        continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    key = n.replace("void ", "").replace("(anonymous namespace)::", "").split(">(")[0][:90]
    by[key].append(d)
for k, v in sorted(by.items()):
    v.sort()
    # clusters: split where consecutive sorted durations differ by > 12 %
    cl, cur = [], [v[0]]
    for x in v[1:]:
        if x > cur[-1] * 1.12:
            cl.append(cur); cur = [x]
        else:
            cur.append(x)
    cl.append(cur)
    print(k, len(v), "launches")
    for c in cl:
        if len(c) >= 3:
            print(f"    {len(c):5d} x  median {c[len(c)//2]:9.1f} us   [{c[0]:.1f} .. {c[-1]:.1f}]")
