O=gpurun_out/r04; mkdir -p $O
MOLLY_ATTN_FWD_WAVES=8 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attn" 2>&1 | tail -3
MOLLY_ATTN_FWD_WAVES=8 python tools/fuzz_attn.py 2>&1 | tail -1
{
for sh in 8,2048,16,8,128 16,2048,16,8,128 16,2048,64,8,128 2,4096,32,8,128 1,3072,32,8,128; do
echo "== $sh   4 waves (two workgroups per CU) / 8 waves (one)"
ATTN_SHAPE=$sh python tools/bench_attn.py 2>&1 | grep "fwd causal"
ATTN_SHAPE=$sh MOLLY_ATTN_FWD_WAVES=8 python tools/bench_attn.py 2>&1 | grep "fwd causal"
done
} | tee $O/attn_fwd8w.log
