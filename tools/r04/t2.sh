cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-secondary --no-vendor-gemm --no-batch8-reference"
for w in 0 256 512 0 256 512; do MOLLY_ROWS_RULE_WIDE=$w python bench.py --steps 6 --warmup 2 $B --model 4b --batch 1 --seq 3072 --micro "dna:512,rna:512,protein:512;dna:512,rna:512,protein:512" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3 wide_rule=$w', d['ms_per_step'])"; done
