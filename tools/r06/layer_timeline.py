#!/usr/bin/env python3
"""The launch sequence of one decoder layer inside the step, from a rocprofv3 --kernel-trace CSV: the last step's launches in start order,
one window from the forward and one from the backward (anchored on the attention kernels), each launch's duration and the gap in front of it;
then the medians per position over every layer of the step.
    python tools/r05/layer_timeline.py <kernel_trace.csv>"""
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:64]


ev = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# the last step: from the last-but-28th attn_fwd (the decoder has 28 layers) on
fw = [i for i, e in enumerate(ev) if e[0].startswith("attn_fwd_kernel<128>") or e[0].startswith("attn_fwd_pipe")]
bw = [i for i, e in enumerate(ev) if e[0].startswith("attn_bwd_dq")]
L = int(sys.argv[2]) if len(sys.argv) > 2 else 28
for name, idx in (("forward", fw[-L:]), ("backward", bw[-L:])):
    # a layer = the launches from one anchor to the next
    segs = [ev[idx[k]:idx[k + 1]] for k in range(len(idx) - 1)]
    n = statistics.mode(len(s) for s in segs)
    segs = [s for s in segs if len(s) == n and all(s[j][0] == segs[0][j][0] for j in range(n))] or segs[:1]
    print(f"== {name}: {n} launches per layer, {len(segs)} layers with the same sequence")
    tot = 0.0
    for j in range(n):
        d = statistics.median((s[j][2] - s[j][1]) / 1e3 for s in segs)
        g = statistics.median((s[j][1] - (s[j - 1][2] if j else s[j][1])) / 1e3 for s in segs)
        tot += d
        print(f"   {j:3d} {segs[0][j][0]:66s} {d:9.1f} us   gap {g:7.1f}")
    span = statistics.median((s[-1][2] - s[0][1]) / 1e3 for s in segs)
    print(f"   sum of durations {tot:9.1f} us; first start -> last end {span:9.1f} us")
