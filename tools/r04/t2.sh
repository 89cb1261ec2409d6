cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_generate.py tests/test_gpu_config5.py tests/test_gpu_fuzz.py tests/test_gpu_edge_cases.py tests/test_gpu_lora.py -q 2>&1 | tail -4
