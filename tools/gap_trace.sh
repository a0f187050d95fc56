cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt; mkdir -p /tmp/kt
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/bench.py --steps 6 --warmup 3 $BENCH_EXTRA --no-cpu-baseline --no-kernels --no-extras > /tmp/kt.log 2>&1
F=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python $R/tools/gap_stats.py $F --window-ms 180
