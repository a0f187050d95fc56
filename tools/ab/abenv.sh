#!/bin/bash
# usage: abenv.sh VAR v1 v2 v1 v2 ...: the default bench with VAR set to each value in turn
V=$1; shift
for v in "$@"; do
  echo "=== $V=$v: $(env $V=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-kernels 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
done
