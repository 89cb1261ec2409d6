#!/bin/bash
# same-box A/B of config 3 (Molly-4B, B = 1, GA = 2) across library variants: $1 = log name, the rest = variant names (base = the in-tree library)
mkdir -p gpurun_out/r05
L=gpurun_out/r05/ab_c3_$1.log; shift
: > $L
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset MOLLY_LIB_PATH; else export MOLLY_LIB_PATH=tools/variants/libmolly_$v.so; fi
    echo "--- variant $v rep $rep" >> $L
    python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary --no-vendor-gemm --no-batch8-reference --model 4b --batch 1 --seq 3072 --micro "dna:512,rna:512,protein:512;dna:512,rna:512,protein:512" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms_p50'], d['value'], d['loss'])" >> $L
  done
done
unset MOLLY_LIB_PATH
cat $L
