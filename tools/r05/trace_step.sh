#!/bin/bash
# kernel trace of the headline step -> the per-layer launch sequence with durations (tools/r05/layer_timeline.py) and per-shape GEMM clusters
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_tr -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --no-vendor-gemm --no-batch8-reference ${EXTRA} > $O/trace_run${1}.log 2>&1
f=$(ls /tmp/p_tr/*/*kernel_trace.csv | head -1)
python3 $R/tools/r05/layer_timeline.py $f > $O/layer_timeline${1}.log
python3 $R/tools/r04/trace_shapes.py $f > $O/trace_shapes${1}.log
cp /tmp/p_tr/*/*kernel_stats.csv $O/step_kernel_stats${1}.csv
rm -rf /tmp/p_tr
cat $O/layer_timeline${1}.log
