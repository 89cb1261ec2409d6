O=gpurun_out/r04; mkdir -p $O
B="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-vendor-gemm"
for b in 8 12 16 24; do
python bench.py $B --batch $b 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('batch $b', d['ms_per_step'], d['value'], d['mfma_roofline_frac_step_executed'], d['roofline']['achieved'])"
done | tee $O/batch_sweep.log
