python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -3
for r in 1 2; do
for v in base nt0 ntl prod; do
  if [ $v = prod ]; then unset MOLLY_LIB_PATH; else export MOLLY_LIB_PATH=$PWD/tools/variants/libmolly_$v.so; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['value'])"
done; done
