R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
python -m pytest tests -x -q -m gpu > $O/gpu_suite_after_decode.log 2>&1; tail -3 $O/gpu_suite_after_decode.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python tools/bench_generate.py --model 1.7b 2>&1 | tail -3 | tee $O/bench_generate_17b.log
python tools/bench_generate.py --model 4b 2>&1 | tail -3 | tee $O/bench_generate_4b.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c5 -- python3 $R/bench.py --secondary-worker c5 > $O/prof_c5.log 2>&1
cp /tmp/p_c5/*/*kernel_stats.csv $R/profiles/r04_c5_8b_generate_kernel_stats.csv
cp $R/profiles/r04_c5_8b_generate_kernel_stats.csv $O/
tail -1 $O/prof_c5.log
