for args in "--train-mode lora" "--train-mode mlp" "--model 4b --batch 4 --seq 3072" "--model 8b --batch 2 --seq 4096 --k-protein 1024" "--batch 1" "--batch 4"; do
  python bench.py $args --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$args', '->', d['ms_per_step'], 'ms', round(d['value']), 'tok/s', d.get('model_tflops_per_gpu'), 'TF/s alg')"
done
python tools/bench_generate.py 2>&1 | tail -4
