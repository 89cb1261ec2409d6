export MOLLY_FUSED_SWIGLU_BWD=0
for r in 1 2 3; do
for v in old prod; do
  if [ $v = prod ]; then unset MOLLY_LIB_PATH; else export MOLLY_LIB_PATH=$PWD/tools/variants/libmolly_$v.so; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['value'], d['roofline']['achieved'])"
done; done
unset MOLLY_LIB_PATH
MOLLY_FUSED_SWIGLU_BWD=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prod fused_bwd=1', d['ms_per_step'], d['value'], d['roofline']['achieved'])"
