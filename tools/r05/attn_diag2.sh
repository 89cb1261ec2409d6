#!/bin/bash
mkdir -p gpurun_out/r05
L=gpurun_out/r05/attn_diag2_${1:-a}.log
: > $L
for st in 1 0; do
  echo "=== fuzz PIPE=8 STAGGER=$st" >> $L
  MOLLY_ATTN_PIPE_STAGGER=$st MOLLY_ATTN_FWD_PIPE=8 timeout 300 python tools/fuzz_attn.py --cases 40 --seed 5 2>&1 | tail -2 >> $L
done
for st in 1 0; do for pr in 0 1; do
  echo "=== bench STAGGER=$st PRIO=$pr" >> $L
  MOLLY_ATTN_PIPE_STAGGER=$st MOLLY_ATTN_PRIO=$pr timeout 300 python tools/r05/bench_attn_pipe.py --pipes 0,8 2>&1 | grep -v amdgpu.ids >> $L
done; done
for c in 1 0; do
  MOLLY_ATTN_FWD_PIPE=8 ATTN_CAUSAL=$c timeout 120 python tools/r05/attn_pipe_stamp.py 2>&1 | grep -v amdgpu.ids >> $L
done
cat $L
