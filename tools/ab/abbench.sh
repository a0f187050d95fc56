#!/bin/bash
for v in "$@"; do
  cp tools/_variants/$v.so ffwm_amd/lib/libffwm_hip.so
  echo "=== variant $v: $(python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-kernels 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
done
