cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_edge_cases.py tests/test_gpu_streamk.py tests/test_gpu_small_split.py tests/test_gpu_dynamic_fetch.py tests/test_gpu_config4.py -q 2>&1 | tail -4
B="--no-cpu-baseline --no-secondary --no-vendor-gemm"
python bench.py $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline', d['ms_per_step'], d['value'], d['config'].get('batch8_reference',{}).get('step_ms_p50'))"
