python tools/gemm_diag/cmp_libs.py old 2>&1 | tail -2
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -2
for r in 1 2 3; do
for v in 0 1; do
  MOLLY_FUSED_SWIGLU_BWD=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused_bwd=$v', d['ms_per_step'], d['value'], d['roofline']['achieved'], ' NT %.1f NN %.1f' % (d['roofline']['by_kernel']['NT gemm256_kernel<false,false>']['avg_launch_us'], d['roofline']['by_kernel']['NN gemm256_kernel<false,true>']['avg_launch_us']))"
done; done
