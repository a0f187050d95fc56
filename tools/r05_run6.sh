#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/run6
mkdir -p $O
for v in 0 1 2 3; do
  echo "=== be_pair_variant=$v" >> $O/pair_ab.txt
  FFWM_OPTS=be_pair_variant=$v timeout 600 python tools/be_bwd_pair_ab.py 2>&1 | grep -v amdgpu.ids >> $O/pair_ab.txt
done
cat $O/pair_ab.txt
