#!/bin/bash
# rocprofv3 kernel trace of one bench workload -> gpurun_out/prof/<tag>_kernel_stats.csv (steady-state tail folded by
# tools/steady_stats.py) + the bench line.   tools/prof_workload.sh <tag> <window-ms> <bench.py args...>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; WIN=$2; shift; shift
OUT=$R/gpurun_out/prof
mkdir -p $OUT; rm -rf /tmp/kt_$TAG; mkdir -p /tmp/kt_$TAG
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$TAG -- python $R/bench.py "$@" --no-cpu-baseline --no-kernels > $OUT/${TAG}_bench.log 2>&1
F=$(find /tmp/kt_$TAG -name "*kernel_trace.csv" | head -1)
python $R/tools/steady_stats.py $F $OUT/${TAG}_kernel_stats.csv --window-ms $WIN --header "rocprofv3 --kernel-trace -- python bench.py $* --no-cpu-baseline --no-kernels; last $WIN ms of the trace (tools/prof_workload.sh)" > $OUT/${TAG}_steady.log 2>&1
tail -1 $OUT/${TAG}_bench.log | cut -c1-400
head -40 $OUT/${TAG}_kernel_stats.csv
