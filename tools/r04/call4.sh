# round 4, call 4: packed-fp32 softmax math in the attention kernels against the committed library; whole-step A/B of a few knobs;
# the two encoders on two streams at the C3 shape
O=gpurun_out/r04; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attn" 2>&1 | tail -2
for i in 1 2; do
echo "== HEAD library";  MOLLY_LIB_PATH=tools/variants/libmolly_head.so python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids
echo "== packed fp32";   python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids
done > $O/attn_pk.log 2>&1; cat $O/attn_pk.log
echo "== one-pass dK+dV (1 wave per SIMD)" >> $O/attn_pk.log; MOLLY_ATTN_DKV_ONE_PASS=1 python tools/bench_attn.py 2>&1 | grep "bwd" | tee -a $O/attn_pk.log
B="--steps 8 --warmup 3 --no-cpu-baseline --no-secondary --no-vendor-gemm"
run() { python bench.py $B "$@" 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['ms_per_step'], d['step_ms_p50'])"; }
for i in 1 2; do
echo "head lib   : $(MOLLY_LIB_PATH=tools/variants/libmolly_head.so run)"
echo "this lib   : $(run)"
echo "small3=1   : $(MOLLY_GEMM_SET=small3=1 run)"
done | tee $O/ab_step.log
C3="--model 4b --batch 1 --seq 3072 --micro dna:512,rna:512,protein:512;dna:512,rna:512,protein:512"
for i in 1 2; do
echo "C3 one stream : $(MOLLY_ENC_STREAMS=0 run $C3)"
echo "C3 two streams: $(run $C3)"
done | tee $O/ab_c3_streams.log
