cd $GRAFT_REPO_ROOT
MOLLY_GEMM_SET="rows_bn=64" timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "other_model_widths" 2>&1 | grep -B5 -A25 "^E " | head -60
