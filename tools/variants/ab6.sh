export MOLLY_FUSED_SWIGLU_BWD=0
for v in old prod old prod; do
  if [ $v = prod ]; then unset MOLLY_LIB_PATH; else export MOLLY_LIB_PATH=$PWD/tools/variants/libmolly_$v.so; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['value'], d['roofline']['achieved'])
for k,v in d['roofline']['by_kernel'].items(): print('    %-60s %5d launches  %8.1f us  %7.1f TF/s' % (k[:60], v['launches'], v['avg_launch_us'], v['achieved_tflops']))
"
done
