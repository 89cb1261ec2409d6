cd $GRAFT_REPO_ROOT
python tools/bench_decode_gemm.py --batch 32 2>&1 | grep "lm_head" | cut -c1-110
MOLLY_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/libmolly_head.so python tools/bench_decode_gemm.py --batch 32 2>&1 | grep "lm_head" | cut -c1-110
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "gemm" 2>&1 | tail -2
sed -i "s/print('batch'/print('lmnt '/" tools/r04/t.sh
bash tools/r04/t.sh
