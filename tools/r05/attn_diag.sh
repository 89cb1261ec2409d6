#!/bin/bash
mkdir -p gpurun_out/r05
L=gpurun_out/r05/attn_diag_${1:-a}.log
: > $L
timeout 300 python tools/r05/bench_attn_pipe.py >> $L 2>&1
for c in 1 0; do
  MOLLY_ATTN_FWD_PIPE=8 ATTN_CAUSAL=$c timeout 120 python tools/r05/attn_pipe_stamp.py >> $L 2>&1
done
timeout 600 bash tools/r05/attn_pmc.sh ${1:-a} >> $L 2>&1
cat $L
