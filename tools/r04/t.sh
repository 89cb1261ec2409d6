cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_streamk.py -q -k "carves or streamk" 2>&1 | tail -12
