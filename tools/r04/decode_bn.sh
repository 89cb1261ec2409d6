O=$GRAFT_REPO_ROOT/gpurun_out/r04; mkdir -p $O
cd $GRAFT_REPO_ROOT
MOLLY_GEMM_SET="rows_bn=64" timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_generate.py -q -k "decode or generate" 2>&1 | tail -5 > $O/decode_bn_tests.log
for m in 64 128 64 128; do MOLLY_GEMM_SET="rows_bn=$m" python bench.py --secondary-worker c5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rows_bn=$m', d['decode']['ms_per_step'], d['decode']['frac'])" | tee -a $O/c5_bn.log; done
