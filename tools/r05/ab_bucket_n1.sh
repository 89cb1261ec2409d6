#!/bin/bash
# N = 1: does the size of the side-stream AdamW launches (= the bucket) matter to the step?  same-box, interleaved
mkdir -p gpurun_out/r05
L=gpurun_out/r05/ab_bucket_n1.log
: > $L
for rep in 1 2; do
  for mib in 0 8 16 64 4; do
    echo "--- bucket MiB $mib (0 = default 32) rep $rep" >> $L
    python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary --no-vendor-gemm --no-batch8-reference --bucket-mib $mib 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms_p50'], d['value'], d['loss'])" >> $L
  done
done
cat $L
