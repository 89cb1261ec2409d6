#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary --no-vendor-gemm --no-batch8-reference"
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_c3 -- python3 $R/bench.py --steps 5 --warmup 2 $B --model 4b --batch 1 --seq 3072 --micro "dna:512,rna:512,protein:512;dna:512,rna:512,protein:512" > $O/c3_gaps_run.log 2>&1
python3 $R/tools/r05/step_gaps.py $(ls /tmp/p_c3/*/*kernel_trace.csv | head -1) > $O/c3_gaps.log 2>&1
cat $O/c3_gaps.log
rm -rf /tmp/p_c3
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_b16 -- python3 $R/bench.py --steps 4 --warmup 2 $B > $O/b16_gaps_run.log 2>&1
python3 $R/tools/r05/step_gaps.py $(ls /tmp/p_b16/*/*kernel_trace.csv | head -1) > $O/b16_gaps.log 2>&1
cat $O/b16_gaps.log
