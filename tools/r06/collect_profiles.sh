# Round-5 profile collection (run on the GPU box through gpurun).  Kernel-trace statistics of the headline step, of BASELINE
# configs 3 / 4 (Molly-4B / 8B, B = 1, GA = 2) and of config 5 (8B prefill + decode); then the counter passes of the headline step, each
# `--pmc` set in a run of its own with --kernel-trace only.  Summaries are written to profiles/r06_* and copied to gpurun_out/r06/.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_step /tmp/p_step8 /tmp/p_c3 /tmp/p_c4 /tmp/p_c5 /tmp/p_sq_a /tmp/p_sq_b /tmp/p_f /tmp/p_w /tmp/p_lora   # (a box may be handed out twice: stale runs would make the globs below ambiguous)
B="--no-cpu-baseline --no-secondary --no-vendor-gemm --no-batch8-reference"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_step -- python3 $R/bench.py --steps 4 --warmup 2 $B > $O/prof_step.log 2>&1
cp /tmp/p_step/*/*kernel_stats.csv $R/profiles/r06_bench_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_step8 -- python3 $R/bench.py --batch 8 --steps 4 --warmup 2 $B > $O/prof_step8.log 2>&1
cp /tmp/p_step8/*/*kernel_stats.csv $R/profiles/r06_bench_b8_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c3 -- python3 $R/bench.py --steps 3 --warmup 2 $B --model 4b --batch 1 --seq 3072 --micro "dna:512,rna:512,protein:512;dna:512,rna:512,protein:512" > $O/prof_c3.log 2>&1
cp /tmp/p_c3/*/*kernel_stats.csv $R/profiles/r06_c3_4b_b1_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c4 -- python3 $R/bench.py --steps 3 --warmup 2 $B --model 8b --batch 1 --seq 4096 --micro "protein:1024;dna:1000" > $O/prof_c4.log 2>&1
cp /tmp/p_c4/*/*kernel_stats.csv $R/profiles/r06_c4_8b_b1_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c5 -- python3 $R/bench.py --secondary-worker c5 > $O/prof_c5.log 2>&1
cp /tmp/p_c5/*/*kernel_stats.csv $R/profiles/r06_c5_8b_generate_kernel_stats.csv
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/p_sq_a -- python3 $R/bench.py --steps 1 --warmup 1 $B > $O/pmc_sq_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /tmp/p_sq_b -- python3 $R/bench.py --steps 1 --warmup 1 $B > $O/pmc_sq_b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_f -- python3 $R/bench.py --steps 1 --warmup 1 $B > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_w -- python3 $R/bench.py --steps 1 --warmup 1 $B > $O/pmc_w.log 2>&1
cd $R
python tools/pmc_sq_summary.py r06 "round 6 final: streaming + whole-line GEMM epilogues, rotary backward inside the attention backward, dK + dV in one pass" /tmp/p_sq_a /tmp/p_sq_b | head -30
python tools/pmc_hbm_traffic.py /tmp/p_f /tmp/p_w r06 "python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary   (MI355X, B=16 T=2048 K=512; 2 steps incl. warmup)" | head -40
cp profiles/r06_* $O/ 2>/dev/null
ls -la $O | head -40
# the LoRA step (secondary.lora) and the per-layer launch timelines of the headline step and of config 3
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_lora -- python3 $R/bench.py --train-mode lora --steps 3 --warmup 1 $B > $O/prof_lora.log 2>&1
cp /tmp/p_lora/*/*kernel_stats.csv $R/profiles/r06_lora_kernel_stats.csv
python3 $R/tools/r06/layer_timeline.py $(ls /tmp/p_step/*/*kernel_trace.csv | head -1) > $R/profiles/r06_layer_timeline_b16.txt 2>&1
python3 $R/tools/r06/layer_timeline.py $(ls /tmp/p_c3/*/*kernel_trace.csv | head -1) 36 > $R/profiles/r06_layer_timeline_c3.txt 2>&1
python3 $R/tools/r04/trace_shapes.py $(ls /tmp/p_step/*/*kernel_trace.csv | head -1) > $R/profiles/r06_trace_shapes_b16.txt 2>&1
cp $R/profiles/r06_* $O/ 2>/dev/null
