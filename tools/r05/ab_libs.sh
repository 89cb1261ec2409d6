#!/bin/bash
# same-box A/B of the headline step across library variants: $1 = log name, the rest = variant names ("" = the in-tree library)
mkdir -p gpurun_out/r05
L=gpurun_out/r05/ab_$1.log; shift
: > $L
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset MOLLY_LIB_PATH; else export MOLLY_LIB_PATH=tools/variants/libmolly_$v.so; fi
    echo "--- variant $v rep $rep" >> $L
    python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary --no-vendor-gemm --no-batch8-reference 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms_p50'], d['value'], d['loss'])" >> $L
  done
done
unset MOLLY_LIB_PATH
cat $L
