mkdir -p gpurun_out/r04
MOLLY_BENCH_DEVICE=0 MOLLY_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r04/bench_2ranks_gloo.json 2> gpurun_out/r04/bench_2ranks_gloo.err; echo rc=$?; tail -c 2500 gpurun_out/r04/bench_2ranks_gloo.json; tail -5 gpurun_out/r04/bench_2ranks_gloo.err
