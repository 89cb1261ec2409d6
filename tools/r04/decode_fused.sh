O=$GRAFT_REPO_ROOT/gpurun_out/r04; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd $R
MOLLY_DECODE_NW=16 python tools/r04/dbg_decode_qkv.py 2>&1 | tail -8 > $O/dbg_decode_qkv.log
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_generate.py tests/test_gpu_config5.py tests/test_gpu_model.py -q 2>&1 | tail -30 > $O/decode_fused_tests.log
python bench.py --secondary-worker c5 2>&1 | tail -1 | tee $O/c5_fused.json
