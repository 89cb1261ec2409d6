cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_prof_v3 -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r02_prof_v3.log 2>&1
tail -1 gpurun_out/r02_prof_v3.log | cut -c1-200
bash tools/collect_pmc.sh > gpurun_out/r02b_pmc.log 2>&1
tail -45 gpurun_out/r02b_pmc.log | head -16
