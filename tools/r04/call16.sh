O=gpurun_out/r04; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/gpu_suite3.log 2>&1; tail -3 $O/gpu_suite3.log
B="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-vendor-gemm --no-batch8-reference"
run() { python bench.py $B "$@" 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['ms_per_step'], d['step_ms_p50'], d['loss'])"; }
for i in 1 2; do
echo "head lib    : $(MOLLY_LIB_PATH=tools/variants/libmolly_head.so run)"
echo "rcp sigmoid : $(run)"
done | tee $O/ab_rcp.log
