mkdir -p gpurun_out/r04; O=gpurun_out/r04
B="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-vendor-gemm --no-batch8-reference"
run() { python bench.py $B "$@" 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['ms_per_step'], d['step_ms_p50'])"; }
for g in 4 8 2 16 4 8; do echo "group_m $g: $(MOLLY_GEMM_SET=group_m=$g run)"; done | tee $O/ab_group_m_b16.log
