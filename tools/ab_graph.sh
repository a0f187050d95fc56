#!/bin/bash
# A/B of environment switches on the default (graph replay) train bench.  usage: ab_graph.sh "<label>=<ENV=VAL ...>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/ab
for spec in "$@"; do
  label=${spec%%=*}; envs=${spec#*=}
  out=$(env $envs python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-kernels --no-extras 2>/dev/null | tail -1)
  echo "$label ($envs): $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms", d["value"], "img/s", d["config"]["launch"])')" | tee -a $R/gpurun_out/ab/results.txt
done
