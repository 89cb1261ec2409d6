mkdir -p gpurun_out/r04
R=$GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/r04/final_gpu_suite.log 2>&1; tail -3 gpurun_out/r04/final_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/r04/t4_bench.json 2> gpurun_out/r04/t4_bench.err; tail -c 300 gpurun_out/r04/t4_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_dec -- python3 $R/bench.py --secondary-worker c5 > $R/gpurun_out/r04/decode_trace_run.log 2>&1
f=$(ls /tmp/p_dec/*/*kernel_trace.csv | head -1)
python3 $R/tools/r04/decode_trace.py $f | tee $R/gpurun_out/r04/decode_trace.log
cp /tmp/p_dec/*/*kernel_stats.csv $R/gpurun_out/r04/r04_c5_8b_generate_kernel_stats.csv
