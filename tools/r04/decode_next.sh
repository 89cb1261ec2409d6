O=$GRAFT_REPO_ROOT/gpurun_out/r04; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_generate.py tests/test_gpu_config5.py -q -k "argmax or generate or config5 or decode" 2>&1 | tail -8 > $O/decode_next_tests.log
python bench.py --secondary-worker c5 2>&1 | tail -1 | tee $O/c5_next.json
python bench.py --secondary-worker c5 2>&1 | tail -1 | tee $O/c5_next2.json
