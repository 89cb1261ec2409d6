O=gpurun_out/r04; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_streamk.py tests/test_gpu_dynamic_fetch.py -x -q -m gpu -k "gemm or swiglu or residual or streamk or dynamic" 2>&1 | tail -2
python tools/gemm_diag/cmp_libs.py head 2>&1 | tail -3
B="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-vendor-gemm --no-batch8-reference"
run() { python bench.py $B "$@" 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['ms_per_step'], d['step_ms_p50'])"; }
for i in 1 2; do
echo "head lib          : $(MOLLY_LIB_PATH=tools/variants/libmolly_head.so run)"
echo "pipelined epilogue: $(run)"
done | tee $O/ab_epilogue.log
