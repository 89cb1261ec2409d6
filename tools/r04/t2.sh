cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-secondary --no-vendor-gemm --no-batch8-reference"
MOLLY_GEMM_TABLE=gpurun_out/r04/c3_gemm_table3.txt python bench.py --steps 2 --warmup 2 --event-stride 1 $B --model 4b --batch 1 --seq 3072 --micro "dna:512,rna:512,protein:512;dna:512,rna:512,protein:512" 2>/dev/null | tail -1 | cut -c1-100
head -3 gpurun_out/r04/c3_gemm_table3.txt; grep "^ *1\.61 \|^ *617\|^ *618\|^ *619" gpurun_out/r04/c3_gemm_table3.txt
