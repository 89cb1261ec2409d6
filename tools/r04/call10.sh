O=gpurun_out/r04; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/gpu_suite2.log 2>&1; tail -3 $O/gpu_suite2.log
python bench.py > $O/t1_bench.json 2> $O/t1_bench.err; tail -c 600 $O/t1_bench.json; tail -3 $O/t1_bench.err
