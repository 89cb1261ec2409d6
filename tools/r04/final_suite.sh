mkdir -p gpurun_out/r04
R=$GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/r04/final_gpu_suite.log 2>&1; tail -2 gpurun_out/r04/final_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > gpurun_out/r04/t7_bench.json 2> gpurun_out/r04/t7_bench.err; tail -c 200 gpurun_out/r04/t7_bench.json
