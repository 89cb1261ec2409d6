mkdir -p gpurun_out/r04; O=gpurun_out/r04
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')"
B="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-vendor-gemm --no-batch8-reference"
run() { python bench.py $B "$@" 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['ms_per_step'], d['step_ms_p50'], d['step_ms'][:3])"; }
for i in 1 2; do
echo "default stream       : $(run)"
echo "high-priority stream : $(MOLLY_BENCH_HIPRIO=1 run)"
done | tee $O/ab_hiprio.log
for i in 1; do
echo "B8 default stream       : $(run --batch 8)"
echo "B8 high-priority stream : $(MOLLY_BENCH_HIPRIO=1 run --batch 8)"
done | tee -a $O/ab_hiprio.log
