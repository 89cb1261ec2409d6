cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/r02b_pmc_sq_a -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r02b_pmc_sq_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/r02b_pmc_sq_b -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r02b_pmc_sq_b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r02b_pmc_f -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r02b_pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r02b_pmc_w -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r02b_pmc_w.log 2>&1
python tools/pmc_sq_summary.py r02b "round 2 final: rolling prefetch, 16-byte regrouped epilogues, fused SwiGLU fwd+bwd" gpurun_out/r02b_pmc_sq_a gpurun_out/r02b_pmc_sq_b | head -20
python tools/pmc_hbm_traffic.py gpurun_out/r02b_pmc_f gpurun_out/r02b_pmc_w r02b | head -30
cp profiles/r02b_* gpurun_out/ 2>/dev/null
du -sh gpurun_out/r02b_pmc_* | tail -5
