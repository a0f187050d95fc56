cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt && mkdir -p /tmp/kt $R/gpurun_out/prof3
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernels > $R/gpurun_out/prof3/bench.log 2>&1
F=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python $R/tools/steady_stats.py $F $R/gpurun_out/prof3/train_step_kernel_stats.csv --window-ms 200 --header "rocprofv3 --kernel-trace -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernels; last 200 ms of the trace"
tail -1 $R/gpurun_out/prof3/bench.log | cut -c1-200
