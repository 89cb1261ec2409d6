cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
MOLLY_GEMM_SET="rows_bn=32" timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_generate.py -q -k "decode or generate" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_generate.py -q -k "decode or generate" 2>&1 | tail -2
for m in 64 32; do echo "== rows_bn=$m"; MOLLY_GEMM_SET="rows_bn=$m" python tools/bench_decode_gemm.py --batch 32 2>&1 | grep -v amdgpu | cut -c1-100; done | tee $O/rows_bn32_bench.log
for m in 32 64 32 64; do MOLLY_GEMM_SET="rows_bn=$m" python bench.py --secondary-worker c5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rows_bn=$m', d['decode']['ms_per_step'], d['decode']['frac'])"; done | tee $O/c5_bn32.log
