O=$GRAFT_REPO_ROOT/gpurun_out/r04; mkdir -p $O
cd $GRAFT_REPO_ROOT
for m in 128 64; do echo "== rows_bn=$m"; MOLLY_GEMM_SET="rows_bn=$m" python tools/bench_decode_gemm.py --batch 32 24 2>&1 | grep -v amdgpu | cut -c1-130; done | tee $O/rows_bn_bench.log
for mdl in 1.7b 4b; do for m in 128 64; do MOLLY_GEMM_SET="rows_bn=$m" python tools/bench_generate.py --model $mdl 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mdl rows_bn=$m', d['decode_ms_per_step'])"; done; done | tee $O/generate_bn.log
for b in 8 20; do for m in 128 64; do MOLLY_GEMM_SET="rows_bn=$m" python tools/bench_generate.py --model 8b --batch $b --new 40 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('8b B=$b rows_bn=$m', d['decode_ms_per_step'])"; done; done | tee -a $O/generate_bn.log
