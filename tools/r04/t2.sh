cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-secondary --no-vendor-gemm --no-batch8-reference"
for w in 1 2 1 2; do MOLLY_GEMM_STREAMK=$w python bench.py --steps 6 --warmup 2 $B --model 4b --batch 1 --seq 3072 --micro "dna:512,rna:512,protein:512;dna:512,rna:512,protein:512" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3 streamk=$w', d['ms_per_step'])"; done
MOLLY_GEMM_STREAMK=2 MOLLY_GEMM_TABLE=gpurun_out/r04/c3_gemm_table_sk2.txt python bench.py --steps 2 --warmup 2 --event-stride 1 $B --model 4b --batch 1 --seq 3072 --micro "dna:512,rna:512,protein:512;dna:512,rna:512,protein:512" 2>/dev/null | tail -1 | cut -c1-100
head -14 gpurun_out/r04/c3_gemm_table_sk2.txt
