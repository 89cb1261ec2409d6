# whole-step A/B of an environment switch on ONE box: alternating processes (bench.py --steps 10, headline workload, nothing else measured)
# usage: bash tools/r06/ab_env_step.sh VAR A B [rounds]
VAR=$1; A=$2; B=$3; N=${4:-2}
O=gpurun_out/r06; mkdir -p $O
for i in $(seq 1 $N); do
  for v in $A $B; do
    env $VAR=$v python bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-vendor-gemm --no-batch8-reference 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$VAR=$v', 'ms_per_step', d['ms_per_step'], 'p50', d.get('step_ms_p50'), 'tokens/s', d['value'])"
  done
done
