O=gpurun_out/r04; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_edge_cases.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -2
python tools/fuzz_attn.py 2>&1 | tail -2
for i in 1 2; do
echo "== HEAD library";  MOLLY_LIB_PATH=tools/variants/libmolly_head.so python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids
echo "== hoisted staging";   python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids
done > $O/attn_stage.log 2>&1; cat $O/attn_stage.log
{ python tools/r04/attn_stamp.py; } 2>&1 | grep -v amdgpu.ids | tee $O/attn_stamp2.log
for s in 1,3072,32,8,128 2,4096,32,8,128; do echo "== $s head / new"; ATTN_SHAPE=$s MOLLY_LIB_PATH=tools/variants/libmolly_head.so python tools/bench_attn.py 2>&1 | grep "fwd causal\|bwd"; ATTN_SHAPE=$s python tools/bench_attn.py 2>&1 | grep "fwd causal\|bwd"; done | tee -a $O/attn_stage.log
