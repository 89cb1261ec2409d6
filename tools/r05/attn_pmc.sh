#!/bin/bash
# SQ counters of the two forward kernels on the same problem (two passes, --kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-a}
D=gpurun_out/r05/attn_pmc_$T
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d ${D}_a -- python3 tools/r05/bench_attn_pipe.py --reps 3 --shapes "16,2048,16,8,128" > ${D}_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d ${D}_b -- python3 tools/r05/bench_attn_pipe.py --reps 3 --shapes "16,2048,16,8,128" > ${D}_b.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("${D}_a", "${D}_b"):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "attn_fwd" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, c in agg.items():
        print(k)
        for n, v in sorted(c.items()):
            print(f"   {n:28s} {v:.4g}")
        if "SQ_BUSY_CU_CYCLES" in c:
            print("   mfma_busy", c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * c["SQ_BUSY_CU_CYCLES"]), "valu/mfma", (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"])
        if "SQ_WAVE_CYCLES" in c:
            w = c["SQ_WAVE_CYCLES"]
            print("   wait", c["SQ_WAIT_ANY"] / w, "stall", c["SQ_WAIT_INST_ANY"] / w, "active", c["SQ_ACTIVE_INST_ANY"] / w, "lds-stall", c.get("SQ_WAIT_INST_LDS", 0) / w)
PY
rm -rf ${D}_a ${D}_b
