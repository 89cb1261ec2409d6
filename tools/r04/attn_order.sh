# round 4, first call: the XCD-aware attention block order (block_item in attention.hip) against round 3's order
O=gpurun_out/r04; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attn" > $O/attn_tests.log 2>&1; tail -3 $O/attn_tests.log
for s in 0 1 2 4; do
  echo "== MOLLY_ATTN_ORDER_SET=$s MOLLY_ATTN_ORDER_SET_DKV=$s"
  MOLLY_ATTN_ORDER_SET=$s MOLLY_ATTN_ORDER_SET_DKV=$s python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids
done > $O/attn_order.log 2>&1
for s in 0 1 2; do
  echo "== B=1 T=3072 32/8 heads  set=$s"
  MOLLY_ATTN_ORDER_SET=$s MOLLY_ATTN_ORDER_SET_DKV=$s ATTN_SHAPE=1,3072,32,8,128 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids | head -2
  echo "== B=2 T=4096 32/8 heads  set=$s"
  MOLLY_ATTN_ORDER_SET=$s MOLLY_ATTN_ORDER_SET_DKV=$s ATTN_SHAPE=2,4096,32,8,128 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids | head -2
done >> $O/attn_order.log 2>&1
cat $O/attn_order.log
for s in 0 1; do
  MOLLY_ATTN_ORDER_SET=$s MOLLY_ATTN_ORDER_SET_DKV=$s python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_order$s.json 2> $O/bench_order$s.err
  python -c "import json;d=json.loads(open('$O/bench_order$s.json').read().strip().splitlines()[-1]);print('order',$s,d['ms_per_step'],d['value'])"
done
