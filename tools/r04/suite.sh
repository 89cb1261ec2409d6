O=gpurun_out/r04; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; tail -5 $O/gpu_suite.log
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_a.json 2> $O/bench_a.err; tail -c 1500 $O/bench_a.json
