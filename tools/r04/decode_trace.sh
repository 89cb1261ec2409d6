O=$GRAFT_REPO_ROOT/gpurun_out/r04; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_dec -- python3 $R/bench.py --secondary-worker c5 > $O/decode_trace_run.log 2>&1
f=$(ls /tmp/p_dec/*/*kernel_trace.csv | head -1)
python3 $R/tools/r04/decode_trace.py $f | tee $O/decode_trace.log
