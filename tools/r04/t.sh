cd $GRAFT_REPO_ROOT
for rep in 1 2; do
python bench.py --secondary-worker c5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch', d['decode']['ms_per_step'], d['decode']['frac'])"
MOLLY_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/libmolly_head.so python bench.py --secondary-worker c5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('head', d['decode']['ms_per_step'], d['decode']['frac'])"
done
