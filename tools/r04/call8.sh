O=gpurun_out/r04; mkdir -p $O
timeout 1700 python -m pytest tests/test_gpu_two_ranks.py tests/test_gpu_generate.py tests/test_gpu_kernels.py -x -q -m gpu -k "four_ranks or rccl_first or launches_its_own or captured_decode or refuses_to_grow or attn" > $O/new_tests.log 2>&1; tail -15 $O/new_tests.log
