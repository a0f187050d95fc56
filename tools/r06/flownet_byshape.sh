#!/bin/bash
# per-launch-geometry kernel table of the lean FlowNet forward (which LAYER costs what inside the replayed graph)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/kt_fl && mkdir -p /tmp/kt_fl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_fl -- python $R/bench.py --workload flownet --steps 40 --warmup 10 --no-kernels --no-cpu-baseline --no-extras > /tmp/fl.log 2>&1
python $R/tools/steady_stats.py $(find /tmp/kt_fl -name "*kernel_trace.csv" | head -1) $R/gpurun_out/flownet_byshape.csv --window-ms 15 --by-shape --header "lean FlowNet forward, last 15 ms, per kernel and launch geometry"
