python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_train.py tests/test_gpu_wide_layer.py -x -q -m gpu 2>&1 | tail -3
for r in 1 2 3; do
for v in 0 1; do
  MOLLY_FUSED_SWIGLU_BWD=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused_bwd=$v', d['ms_per_step'], d['value'])"
done; done
