cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_generate.py tests/test_gpu_config5.py tests/test_gpu_fuzz.py tests/test_gpu_edge_cases.py tests/test_gpu_streamk.py tests/test_gpu_small_split.py -q 2>&1 | tail -5
