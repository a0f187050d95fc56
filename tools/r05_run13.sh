#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/run13
mkdir -p $O
timeout 600 python tools/warp_feat_gps_sweep.py 2>&1 | grep -v amdgpu.ids > $O/warp_feat_gps.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "warp" > $O/pytest_warp.txt 2>&1; echo "rc $?" >> $O/pytest_warp.txt
cat $O/warp_feat_gps.txt; tail -n 5 $O/pytest_warp.txt
