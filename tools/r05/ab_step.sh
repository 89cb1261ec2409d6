#!/bin/bash
# same-box A/B of the headline step: "$1" = env assignments of arm B (arm A = defaults), e.g. "MOLLY_NORM_TRANSPOSED_STORE=0"
mkdir -p gpurun_out/r05
L=gpurun_out/r05/ab_${2:-x}.log
: > $L
for rep in 1 2; do
  for arm in A B; do
    if [ $arm = B ]; then E="$1"; else E=""; fi
    echo "--- arm $arm ($E) rep $rep" >> $L
    env $E python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary --no-vendor-gemm --no-batch8-reference 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms_p50'], d['value'], d['loss'])" >> $L
  done
done
cat $L
