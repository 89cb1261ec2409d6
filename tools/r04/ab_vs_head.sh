# whole-step / kernel A/B of the working tree against the committed library (python tools/build_head_variant.py first)
O=gpurun_out/r04; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_streamk.py tests/test_gpu_dynamic_fetch.py tests/test_gpu_small_split.py -x -q -m gpu -k "gemm or streamk or dynamic or split" 2>&1 | tail -2
python tools/gemm_diag/cmp_libs.py head 2>&1 | tail -2
python tools/gemm_diag/run_kscan.py head 2>&1 | grep -v amdgpu.ids | tee $O/kscan_decode.log
B="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-vendor-gemm --no-batch8-reference"
run() { python bench.py $B "$@" 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['ms_per_step'], d['step_ms_p50'], d['loss'])"; }
for i in 1 2 3; do
echo "head lib     : $(MOLLY_LIB_PATH=tools/variants/libmolly_head.so run)"
echo "cheap decode : $(run)"
done | tee $O/ab_decode.log
