export MOLLY_FUSED_SWIGLU_BWD=0
for r in 1 2 3; do
for v in head prod; do
  if [ $v = prod ]; then unset MOLLY_LIB_PATH; else export MOLLY_LIB_PATH=$PWD/tools/variants/libmolly_$v.so; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['value'], d['roofline']['achieved'], ' NT %.1f NN %.1f' % (d['roofline']['by_kernel']['NT gemm256_kernel<false,false>']['avg_launch_us'], d['roofline']['by_kernel']['NN gemm256_kernel<false,true>']['avg_launch_us']))"
done; done
