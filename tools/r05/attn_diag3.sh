#!/bin/bash
mkdir -p gpurun_out/r05
L=gpurun_out/r05/attn_diag3_${1:-a}.log
: > $L
PIPES=${PIPES:-0,8}
echo "=== fuzz" >> $L
for nw in ${FUZZ_NW:-8}; do MOLLY_ATTN_FWD_PIPE=$nw timeout 300 python tools/fuzz_attn.py --cases 40 --seed 7 2>&1 | tail -1 >> $L; done
for st in ${STAGGERS:-1 0}; do
  echo "=== bench STAGGER=$st" >> $L
  MOLLY_ATTN_PIPE_STAGGER=$st timeout 300 python tools/r05/bench_attn_pipe.py --pipes $PIPES 2>&1 | grep -v amdgpu.ids >> $L
done
for nw in ${STAMP_NW:-8}; do for c in 1 0; do
  MOLLY_ATTN_FWD_PIPE=$nw ATTN_CAUSAL=$c timeout 120 python tools/r05/attn_pipe_stamp.py 2>&1 | grep -v amdgpu.ids >> $L
done; done
cat $L
