O=gpurun_out/r04; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_wide_layer.py tests/test_gpu_train_bio.py tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -3
B="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-vendor-gemm --no-batch8-reference"
run() { python bench.py $B "$@" 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['ms_per_step'], d['step_ms_p50'], d['loss'])"; }
for i in 1 2; do
echo "head lib (8-byte rope bwd): $(MOLLY_LIB_PATH=tools/variants/libmolly_head.so run)"
echo "16-byte rope bwd          : $(run)"
done | tee $O/ab_rope_bwd.log
