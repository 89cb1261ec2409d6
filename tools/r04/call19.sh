O=gpurun_out/r04; mkdir -p $O
MOLLY_ATTN_FWD2=1 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attn" 2>&1 | tail -3
MOLLY_ATTN_FWD2=1 python tools/fuzz_attn.py 2>&1 | tail -2
{
echo "== product"; python tools/bench_attn.py 2>&1 | grep "fwd causal"
echo "== two streams per wave"; MOLLY_ATTN_FWD2=1 python tools/bench_attn.py 2>&1 | grep "fwd causal"
echo "== product B16"; ATTN_SHAPE=16,2048,16,8,128 python tools/bench_attn.py 2>&1 | grep "fwd causal"
echo "== two streams B16"; ATTN_SHAPE=16,2048,16,8,128 MOLLY_ATTN_FWD2=1 python tools/bench_attn.py 2>&1 | grep "fwd causal"
echo "== product B2 T4096 32/8"; ATTN_SHAPE=2,4096,32,8,128 python tools/bench_attn.py 2>&1 | grep "fwd causal"
echo "== two streams"; ATTN_SHAPE=2,4096,32,8,128 MOLLY_ATTN_FWD2=1 python tools/bench_attn.py 2>&1 | grep "fwd causal"
} | tee $O/attn_fwd2.log
