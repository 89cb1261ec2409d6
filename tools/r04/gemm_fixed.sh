# round 4, call 2: where a 256x256 tile's fixed cost goes now (stamps, no-store ablation), attention order sets
O=gpurun_out/r04; mkdir -p $O
python tools/gemm_diag/run_kscan.py nostore > $O/kscan_nostore.log 2>&1; cat $O/kscan_nostore.log | grep -v amdgpu.ids
python tools/gemm_diag/run_tilestamp.py tilestamp > $O/tilestamp.log 2>&1; grep -v amdgpu.ids $O/tilestamp.log
python tools/gemm_diag/run_tilestamp.py stamp_nostore > $O/tilestamp_nostore.log 2>&1; grep -v amdgpu.ids $O/tilestamp_nostore.log
for s in 0 4 8 auto; do
  echo "== attention order set=$s"
  if [ $s = auto ]; then python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids; else
  MOLLY_ATTN_ORDER_SET=$s MOLLY_ATTN_ORDER_SET_DKV=$s python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids; fi
done > $O/attn_order2.log 2>&1; cat $O/attn_order2.log
python tools/bench_gemm.py --torch > $O/gemm_vs_torch_0.log 2>&1; grep -v amdgpu.ids $O/gemm_vs_torch_0.log | tail -40
