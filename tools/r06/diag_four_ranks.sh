# the 4-rank one-GPU rehearsal of bench.py with a watchdog that dumps the ranks' stacks: bash tools/r06/diag_four_ranks.sh [VAR=VALUE ...]
for kv in "$@"; do export "$kv"; done
MOLLY_BENCH_WATCHDOG_S=${WD:-200} MOLLY_BENCH_DEVICE=0 MOLLY_DIST_BACKEND=gloo timeout 400 python bench.py --gpus 4 --model 0.6b --steps 2 --warmup 1 --batch 2 --seq 1024 --k-protein 256 --exposed-comm-steps 0 --gemm-mode-ab-steps 1 ${EXTRA:---bucket-ab-mib 64,1024 --bucket-ab-steps 1 --tune-budget-s 0} > /tmp/o.txt 2> /tmp/e.txt
echo rc $?
tail -c 150 /tmp/o.txt
echo; echo timeouts: $(grep -c "Timeout" /tmp/e.txt)
grep "sweep:" /tmp/e.txt | tail -12
grep -v Gloo /tmp/e.txt | grep -A16 Timeout | grep "File \"/root/repo" | sort | uniq -c | sort -rn | head -12
