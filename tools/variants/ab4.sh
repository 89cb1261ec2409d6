python -m pytest tests/test_gpu_config4.py -x -q -m gpu 2>&1 | grep -E "^E |passed|failed" | head -12
for v in old prod; do
  if [ $v = prod ]; then unset MOLLY_LIB_PATH; else export MOLLY_LIB_PATH=$PWD/tools/variants/libmolly_$v.so; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['value'], d['roofline']['achieved'])
r=d['roofline']
for k in r: 
    if k not in ('bound','achieved','peak','unit','frac','traffic'): print('   ',k, json.dumps(r[k])[:600])
"
done
