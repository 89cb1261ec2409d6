cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_generate.py tests/test_gpu_config5.py -q -k "decode or generate or config5" 2>&1 | tail -3
bash tools/r04/t.sh
