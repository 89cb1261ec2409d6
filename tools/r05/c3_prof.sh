#!/bin/bash
# BASELINE config 3 (Molly-4B, B = 1, GA = 2, three 512-token spans): bench line + kernel stats + GEMM duration clusters
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary --no-vendor-gemm --no-batch8-reference"
C3='--model 4b --batch 1 --seq 3072 --micro dna:512,rna:512,protein:512;dna:512,rna:512,protein:512'
python3 $R/bench.py --steps 6 --warmup 3 $B $C3 > $O/c3_bench${1}.json 2> $O/c3_bench${1}.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c3 -- python3 $R/bench.py --steps 5 --warmup 2 $B $C3 > $O/c3_prof${1}.log 2>&1
cp /tmp/p_c3/*/*kernel_stats.csv $O/c3_kernel_stats${1}.csv
python3 $R/tools/r04/trace_shapes.py $(ls /tmp/p_c3/*/*kernel_trace.csv | head -1) > $O/c3_trace_shapes${1}.log
python3 $R/tools/r05/layer_timeline.py $(ls /tmp/p_c3/*/*kernel_trace.csv | head -1) 36 > $O/c3_layer_timeline${1}.log 2>&1
rm -rf /tmp/p_c3
python3 -c "import json; d=json.loads(open('$O/c3_bench${1}.json').read().strip().splitlines()[-1]); print('c3', d['ms_per_step'], d['value'], d['roofline']['frac'])"
head -30 $O/c3_kernel_stats${1}.csv | cut -c1-180
cat $O/c3_layer_timeline${1}.log
