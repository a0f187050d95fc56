#!/bin/bash
# tools/ab_env.sh VAR v1 v2 ... : the default train bench once per value of an environment variable (A/B of a routing switch)
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-kernels --no-extras 2>/dev/null | V="$VAR=$v" python -c "import sys,json,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(os.environ['V'], d['value'], d['ms_per_step'])"
done
