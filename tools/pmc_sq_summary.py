#!/usr/bin/env python3
"""Aggregate rocprofv3 SQ counter passes into profiles/r<NN>_pmc_sq_summary.csv: per kernel, matrix-pipe busy fraction, vector
instructions per MFMA, LDS instructions per MFMA, LDS bank-conflict cycles and the share of wave time parked in waits.

Collect (each pass on its own, with --kernel-trace only: MI355X_MICROARCH.md §rocprofv3 PMC slots — 8 SQ slots per pass):
    cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace \\
        --output-format csv -d gpurun_out/pmc_sq_a -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace \\
        --output-format csv -d gpurun_out/pmc_sq_b -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    python tools/pmc_sq_summary.py r02 "<note>" gpurun_out/pmc_sq_a gpurun_out/pmc_sq_b

mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)   (4 SIMDs per CU; the counter counts cycles, 32 per 32x32x16 bf16 MFMA)
valu_per_mfma = (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA   (SQ_INSTS_VALU includes the MFMAs)
wait_share = SQ_WAIT_ANY / SQ_WAVE_CYCLES (wave parked at s_waitcnt / barrier), stall_share = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES."""
import csv
import glob
import os
import sys
from collections import defaultdict


def load(d):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert f, f"no counter_collection.csv under {d}"
    agg = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        launches[k].add(row["Dispatch_Id"])
    return agg, {k: len(v) for k, v in launches.items()}


def main():
    tag, note, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    tot = defaultdict(lambda: defaultdict(float))
    nl = {}
    for d in dirs:
        a, n = load(d)
        for k, c in a.items():
            for name, v in c.items():
                tot[k][name] = v                      # one pass per counter set: no double counting
        nl.update(n)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "profiles", f"{tag}_pmc_sq_summary.csv")
    rows = []
    for k, c in tot.items():
        mf = c.get("SQ_INSTS_MFMA", 0.0)
        if mf <= 0:
            continue
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(4.0 * c.get("SQ_BUSY_CU_CYCLES", 0.0), 1.0)
        wc = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        rows.append((c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), k, nl.get(k, 0), busy, (c.get("SQ_INSTS_VALU", 0.0) - mf) / mf,
                     c.get("SQ_INSTS_LDS", 0.0) / mf, c.get("SQ_LDS_BANK_CONFLICT", 0.0), c.get("SQ_WAIT_ANY", 0.0) / wc,
                     c.get("SQ_WAIT_INST_ANY", 0.0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc))
    rows.sort(reverse=True)
    with open(out, "w") as f:
        f.write(f"# rocprofv3 --pmc <SQ counters> --kernel-trace (two passes) -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline   (MI355X; {note})\n")
        f.write("# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES); valu_per_mfma = (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA; "
                "wait/stall/active shares of SQ_WAVE_CYCLES (parked at waitcnt or barrier / issue stall / issuing)\n")
        f.write("kernel,launches,mfma_busy,valu_per_mfma,lds_insts_per_mfma,lds_bank_conflict_cycles,wait_share,stall_share,active_share\n")
        for _, k, n, busy, vpm, lpm, bc, w, s, a in rows[:24]:
            f.write(f"\"{k[:100]}\",{n},{busy:.3f},{vpm:.2f},{lpm:.2f},{int(bc)},{w:.3f},{s:.3f},{a:.3f}\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
