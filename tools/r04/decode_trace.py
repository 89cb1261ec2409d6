#!/usr/bin/env python3
"""Launch-by-launch record of ONE decode step (the captured graph) from a rocprofv3 --kernel-trace CSV of `bench.py --secondary-worker c5`:
duration of every launch and the idle gap in front of it, medians over the last 20 replays.
    python tools/r04/decode_trace.py <kernel_trace.csv>"""
import collections
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:48]


K = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
starts = [i for i, k in enumerate(K) if k[0].startswith("copy_rows_kernel")]
steps = [K[a:b] for a, b in zip(starts[:-1], starts[1:])]
cnt = collections.Counter(len(s) for s in steps)
n_modal = max((c, n) for n, c in cnt.items() if n > 100)[1]
steps = [s for s in steps if len(s) == n_modal][-21:-1]
print(f"{len(steps)} steps of {n_modal} launches")
dur = [[(s[i][2] - s[i][1]) / 1e3 for s in steps] for i in range(n_modal)]
gap = [[(s[i][1] - s[i - 1][2]) / 1e3 if i else 0.0 for s in steps] for i in range(n_modal)]
med = statistics.median
names = [steps[0][i][0] for i in range(n_modal)]
# the launches of one layer in the middle of the stack
att = [i for i, n in enumerate(names) if n.startswith("attn_decode_kernel")]
per = att[1] - att[0]
a = att[len(att) // 2]
lo = a - (a - att[0]) % per
print(f"layer pattern ({per} launches), launch: duration us / gap in front us")
first = att[len(att) // 2] - (att[0] - 0) + 0
base = att[len(att) // 2] - (att[0] - 1)          # the launch after the embedding copy in layer 0 maps to this one
for i in range(base, base + per):
    print(f"  {names[i]:50s} {med(dur[i]):8.2f} {med(gap[i]):8.2f}")
tot_d = collections.defaultdict(float)
tot_g = collections.defaultdict(float)
for i in range(n_modal):
    tot_d[names[i]] += med(dur[i])
    tot_g[names[i]] += med(gap[i])
print("whole step by kernel: sum of durations / sum of gaps in front (us)")
for n in sorted(tot_d, key=lambda n: -tot_d[n]):
    print(f"  {n:50s} {tot_d[n]:9.1f} {tot_g[n]:9.1f}")
print(f"  total {sum(tot_d.values()):.1f} + {sum(tot_g.values()):.1f} us;  step wall {med([(s[-1][2] - s[0][1]) / 1e3 for s in steps]):.1f} us")
