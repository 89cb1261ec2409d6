O=gpurun_out/r04; mkdir -p $O
MOLLY_TEST_GEMM_BLOCKS=-3 python -m pytest tests -x -q -m gpu > $O/gpu_suite_blocks-3.log 2>&1; tail -2 $O/gpu_suite_blocks-3.log
MOLLY_TEST_GEMM_BLOCKS=dyn python -m pytest tests -x -q -m gpu > $O/gpu_suite_dyn.log 2>&1; tail -2 $O/gpu_suite_dyn.log
