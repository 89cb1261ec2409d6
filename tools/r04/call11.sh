O=gpurun_out/r04; mkdir -p $O
python tools/r04/gemm_cold.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm_cold.log
