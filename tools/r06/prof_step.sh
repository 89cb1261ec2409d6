# kernel-trace statistics of the headline step under an environment switch: bash tools/r06/prof_step.sh TAG [VAR=VALUE]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_$TAG
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$TAG -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --no-vendor-gemm --no-batch8-reference > $O/prof_$TAG.log 2>&1
cp /tmp/p_$TAG/*/*kernel_stats.csv $O/${TAG}_kernel_stats.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/${TAG}_kernel_stats.csv")))
for r in rows[:22]:
    n = r["Name"]
    n = n.replace("(anonymous namespace)::", "").split("(")[0][:70]
    print(f"{n:72s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):5.2f} %")
PY
