# round 4, call 5: what two waves per SIMD are worth in the attention kernels, and whether a priority difference puts them in opposite segments
O=gpurun_out/r04; mkdir -p $O
{
echo "== default";                         python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids
echo "== one workgroup per CU (LDS pad)";  MOLLY_ATTN_LDS_PAD=40960 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids
for p in 1 2 3; do echo "== MOLLY_ATTN_PRIO=$p"; MOLLY_ATTN_PRIO=$p python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids; done
echo "== per-segment priority flips (forward)"; MOLLY_LIB_PATH=tools/variants/libmolly_phaseprio.so python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids
echo "== default again";                   python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids
} > $O/attn_prio.log 2>&1; cat $O/attn_prio.log
