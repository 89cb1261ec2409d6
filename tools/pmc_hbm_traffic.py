#!/usr/bin/env python3
"""Aggregate two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; each collected on its own with --kernel-trace, as
MI355X_MICROARCH.md §HBM prescribes) into profiles/r<NN>_pmc_hbm_traffic.csv and profiles/r<NN>_hbm_traffic.json
(what bench.py quotes as roofline.traffic).

    cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    python tools/pmc_hbm_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w r01
hbm bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE reports half the bytes of 16-B/lane streams;
Infinity-Cache hits are included (fabric-side counter)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert f, f"no counter_collection.csv under {d}"
    agg = defaultdict(lambda: [0, 0.0])
    seen = set()
    for row in csv.DictReader(open(f[0])):
        if row["Counter_Name"] != counter:
            continue
        key = row["Kernel_Name"]
        agg[key][1] += float(row["Counter_Value"])
        did = row["Dispatch_Id"]
        if did not in seen:
            seen.add(did)
            agg[key][0] += 1
    return agg


def main():
    df, dw, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    what = sys.argv[4] if len(sys.argv) > 4 else "python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline   (MI355X, B=8 T=2048 K=512; 2 steps incl. warmup)"
    fe, wr = load(df, "FETCH_SIZE"), load(dw, "WRITE_SIZE")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rows, by, tot_b, tot_n = [], {}, 0.0, 0
    for k, (n, fsum) in sorted(fe.items(), key=lambda kv: -kv[1][1]):
        wn, wsum = wr.get(k, (n, 0.0))
        per = (2 * fsum / n + wsum / max(wn, 1)) * 1024
        rows.append((k, n, fsum / n, wsum / max(wn, 1), per))
        if "gemm256_kernel" in k or "gemm_kernel" in k:
            by[k[k.index("gemm"):][:60]] = int(per)
            tot_b += per * n
            tot_n += n
    with open(os.path.join(root, "profiles", f"{tag}_pmc_hbm_traffic.csv"), "w") as f:
        f.write(f"# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, each with --kernel-trace) -- {what}\n"
                "# hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE reads half the bytes of 16-B/lane streams on "
                "gfx950 (MI355X_MICROARCH.md §HBM); Infinity-Cache hits are included (fabric-side counter)\n"
                "kernel,launches,fetch_size_kb_raw_per_launch,write_size_kb_per_launch,hbm_bytes_per_launch\n")
        for k, n, a, b, per in rows[:40]:
            f.write(f"\"{k[:110]}\",{n},{a:.1f},{b:.1f},{int(per)}\n")
    js = {"source": f"profiles/{tag}_pmc_hbm_traffic.csv", "gemm_launches": tot_n,
          "gemm_hbm_bytes_per_launch": int(tot_b / max(tot_n, 1)), "by_kernel": by}
    with open(os.path.join(root, "profiles", f"{tag}_hbm_traffic.json"), "w") as f:
        json.dump(js, f, indent=1)
    print(json.dumps(js, indent=1))


if __name__ == "__main__":
    main()
