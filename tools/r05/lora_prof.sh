#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python3 bench.py --train-mode lora --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --no-vendor-gemm > gpurun_out/r05/lora_bench_${1:-a}.json 2> gpurun_out/r05/lora_bench_${1:-a}.err
tail -c 600 gpurun_out/r05/lora_bench_${1:-a}.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05/lora_prof_${1:-a} -- python3 bench.py --train-mode lora --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-vendor-gemm > gpurun_out/r05/lora_prof_${1:-a}.log 2>&1
f=$(find gpurun_out/r05/lora_prof_${1:-a} -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/r05/lora_kernel_stats_${1:-a}.csv
rm -rf gpurun_out/r05/lora_prof_${1:-a}
head -25 gpurun_out/r05/lora_kernel_stats_${1:-a}.csv | cut -c1-150
