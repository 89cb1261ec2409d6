# names + launch geometry of the hipBLASLt kernels torch.matmul picks for the step's forward shapes (orientation only)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/vk
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vk -o vk -- python3 $R/tools/r06/vendor_kernels.py ${1:-32768} > $O/vendor_kernels.log 2>&1
find /tmp/vk -name "*kernel_stats.csv" -exec cp {} $O/vendor_kernel_stats.csv \;
python3 - > $O/vendor_kernel_names.txt <<PY
import csv, glob
fs = glob.glob("/tmp/vk/**/*kernel_trace.csv", recursive=True)
seen = set()
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"]
    if k in seen:
        continue
    seen.add(k)
    print({x: r[x] for x in r if x in ("LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Workgroup_Size_X", "Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z", "Scratch_Size", "Workgroup_Size", "Grid_Size")})
    print(k[:900])
PY
grep -v "^W2026\|^E2026" $O/vendor_kernels.log | tail -6
cat $O/vendor_kernel_names.txt | cut -c1-700
