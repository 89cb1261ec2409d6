#!/usr/bin/env python3
"""GPU idle time inside one step, from a rocprofv3 --kernel-trace CSV: the union of all kernel intervals between two consecutive launches of an anchor kernel
(default sqnorm_part_kernel: once per step), the idle remainder, and the largest gaps with the kernels on either side.
    python tools/r05/step_gaps.py <kernel_trace.csv> [anchor substring]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "sqnorm_part_kernel"
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]) for r in rows)
a = [i for i, e in enumerate(ev) if anchor in e[2]]
i0, i1 = a[-2], a[-1]
seg = ev[i0:i1]
t0, t1 = seg[0][0], seg[-1][1]
busy, cur_end, gaps = 0, seg[0][0], []
last = seg[0][2]
for s, e, n in seg:
    if s > cur_end:
        gaps.append((s - cur_end, last, n))
        busy += 0
        cur_start = s
    busy += max(0, e - max(s, cur_end))
    if e > cur_end:
        cur_end, last = e, n
print(f"step span {(t1 - t0) / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle {(t1 - t0 - busy) / 1e6:.2f} ms in {len(gaps)} gaps; {len(seg)} launches")
hist = {}
for g, a_, b_ in gaps:
    k = (a_, b_)
    hist.setdefault(k, [0, 0])
    hist[k][0] += 1; hist[k][1] += g
for (a_, b_), (c, tot) in sorted(hist.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {tot / 1e3:9.1f} us in {c:5d} gaps   after {a_:55s} before {b_}")
