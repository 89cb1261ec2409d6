#!/bin/bash
# is the forward bound by instruction issue?  timing-only variants of attn_fwd_kernel with vector instructions removed
mkdir -p gpurun_out/r05
L=gpurun_out/r05/attn_issue_diag.log
: > $L
for v in ${VARIANTS:-"" attndiag1 attndiag2 attndiag3 attndiag4 ""}; do
  echo "=== variant '$v'" >> $L
  if [ -n "$v" ]; then export MOLLY_LIB_PATH=tools/variants/libmolly_$v.so; else unset MOLLY_LIB_PATH; fi
  timeout 200 python tools/r05/bench_attn_pipe.py --pipes 0 --shapes "16,2048,16,8,128" 2>&1 | grep shape >> $L
done
cat $L
