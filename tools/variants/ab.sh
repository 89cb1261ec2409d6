python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -2
for r in 1 2 3; do
for v in head prod; do
  if [ $v = prod ]; then unset MOLLY_LIB_PATH; else export MOLLY_LIB_PATH=$PWD/tools/variants/libmolly_$v.so; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); bk=d['roofline']['by_kernel']; print('$v', d['ms_per_step'], d['value'], d['roofline']['achieved'], ' '.join('%s %.1f' % (k[:2] + k[-12:], v['avg_launch_us']) for k, v in bk.items() if 'grouped' in k or 'NN gemm256_kernel<false,true>' == k))"
done; done
