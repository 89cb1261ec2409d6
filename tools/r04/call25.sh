mkdir -p gpurun_out/r04
bash tools/r04/collect_profiles.sh > gpurun_out/r04/collect2.log 2>&1; tail -5 gpurun_out/r04/collect2.log
python bench.py > gpurun_out/r04/t2_bench.json 2> gpurun_out/r04/t2_bench.err; tail -c 400 gpurun_out/r04/t2_bench.json
