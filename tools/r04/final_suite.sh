mkdir -p gpurun_out/r04
R=$GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/r04/final_gpu_suite.log 2>&1; tail -2 gpurun_out/r04/final_gpu_suite.log
MOLLY_TEST_GEMM_BLOCKS=-3 python -m pytest tests -x -q -m gpu > gpurun_out/r04/final_gpu_suite_m3.log 2>&1; tail -2 gpurun_out/r04/final_gpu_suite_m3.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > gpurun_out/r04/t5_bench.json 2> gpurun_out/r04/t5_bench.err; tail -c 200 gpurun_out/r04/t5_bench.json
