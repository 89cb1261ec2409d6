#!/bin/bash
# round 5: the software-pipelined attention forward (MOLLY_ATTN_FWD_PIPE = 4 | 8) against attn_fwd_kernel: fuzz, unit tests, timings
mkdir -p gpurun_out/r05
L=gpurun_out/r05/attn_pipe_${1:-a}.log
: > $L
for nw in 8 4; do
  echo "=== fuzz PIPE=$nw" >> $L
  MOLLY_ATTN_FWD_PIPE=$nw timeout 300 python tools/fuzz_attn.py --cases 50 --seed 3 >> $L 2>&1
  echo "rc=$?" >> $L
done
echo "=== pytest attn PIPE=8" >> $L
MOLLY_ATTN_FWD_PIPE=8 timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attn_fwd or attn_bwd" >> $L 2>&1
for shape in 8,2048,16,8,128 16,2048,16,8,128 2,4096,32,8,128 16,2048,64,8,128; do
  for nw in 0 8 4; do
    echo "=== bench shape=$shape PIPE=$nw" >> $L
    ATTN_SHAPE=$shape MOLLY_ATTN_FWD_PIPE=$nw timeout 120 python tools/bench_attn.py 2>&1 | grep "attn fwd" >> $L
  done
done
tail -60 $L
