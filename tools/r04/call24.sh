O=gpurun_out/r04; mkdir -p $O
MOLLY_ATTN_REGSTAGE=1 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attn" 2>&1 | tail -2
MOLLY_ATTN_REGSTAGE=1 python tools/fuzz_attn.py 2>&1 | tail -1
{
for sh in 8,2048,16,8,128 16,2048,16,8,128 16,2048,64,8,128 2,4096,32,8,128; do
echo "== $sh   LDS-DMA / register staging"
ATTN_SHAPE=$sh python tools/bench_attn.py 2>&1 | grep "fwd causal"
ATTN_SHAPE=$sh MOLLY_ATTN_REGSTAGE=1 python tools/bench_attn.py 2>&1 | grep "fwd causal"
done
} | tee $O/attn_regstage.log
