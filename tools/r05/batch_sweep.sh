#!/bin/bash
mkdir -p gpurun_out/r05
L=gpurun_out/r05/batch_sweep.log
: > $L
for b in 16 24 32 16; do
  echo "--- batch $b" >> $L
  python3 bench.py --batch $b --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-vendor-gemm --no-batch8-reference 2>gpurun_out/r05/batch_sweep_$b.err | python3 -c "import sys,json,torch; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms_p50'], d['value'], d['mfma_roofline_frac_step_executed'], d['roofline']['achieved'])" >> $L
done
cat $L
