#!/bin/bash
# rocprofv3 kernel trace of the default train bench -> per-step accounting (tools/step_trace.py).  usage: trace_step.sh <tag> [env...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
rm -rf /tmp/kt_$TAG && mkdir -p /tmp/kt_$TAG $R/gpurun_out/traces
env "$@" timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$TAG -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernels --no-extras > $R/gpurun_out/traces/${TAG}_bench.log 2>&1
F=$(find /tmp/kt_$TAG -name "*kernel_trace.csv" | head -1)
python $R/tools/step_trace.py $F --steps 3 --top 45 > $R/gpurun_out/traces/${TAG}_steps.txt 2>&1
head -24 $R/gpurun_out/traces/${TAG}_steps.txt
