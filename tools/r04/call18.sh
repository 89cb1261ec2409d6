O=gpurun_out/r04; mkdir -p $O
B="--steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-vendor-gemm --no-batch8-reference"
for i in 1 2 3; do python bench.py $B 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('run $i', d['ms_per_step'], repr(d['loss']))"; done | tee $O/bench_repro.log
