O=$GRAFT_REPO_ROOT/gpurun_out/r04; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_tr -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --no-vendor-gemm --no-batch8-reference > $O/trace_run.log 2>&1
f=$(ls /tmp/p_tr/*/*kernel_trace.csv | head -1)
python3 $R/tools/r04/trace_shapes.py $f | tee $O/trace_shapes_b16.log
rm -rf /tmp/p_tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_tr -- python3 $R/bench.py --batch 8 --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --no-vendor-gemm --no-batch8-reference > $O/trace_run8.log 2>&1
f=$(ls /tmp/p_tr/*/*kernel_trace.csv | head -1)
python3 $R/tools/r04/trace_shapes.py $f | tee $O/trace_shapes_b8.log
