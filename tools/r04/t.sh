cd $GRAFT_REPO_ROOT
python tools/r04/bench_rows_gu.py 2>&1 | grep -v amdgpu | tee gpurun_out/r04/rows_gu_bench2.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "decode_rows" 2>&1 | tail -5
for m in 32 1; do MOLLY_GEMM_SET="rows_gu=$m" python bench.py --secondary-worker c5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rows_gu=$m', d['decode']['ms_per_step'], d['decode']['frac'])"; done
