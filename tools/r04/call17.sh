O=gpurun_out/r04; mkdir -p $O
python tools/r04/check_step_determinism.py 16 2>&1 | grep -v amdgpu.ids | tee $O/step_determinism16.log
