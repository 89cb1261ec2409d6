for r in 1 2 3; do
for v in old prod; do
  if [ $v = prod ]; then unset MOLLY_LIB_PATH; else export MOLLY_LIB_PATH=$PWD/tools/variants/libmolly_$v.so; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['value'], d['roofline']['achieved'])"
done; done
unset MOLLY_LIB_PATH
python -m pytest tests -x -q -m gpu 2>&1 | tail -4
