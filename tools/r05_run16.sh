#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/run16
mkdir -p $O
for kf in 0 1; do
  echo "=== conv_fwd_kfast=$kf (batch 8)" >> $O/conv_layers.txt
  B=8 FFWM_OPTS=conv_fwd_kfast=$kf timeout 600 python tools/conv_layers.py 2>&1 | grep -v amdgpu.ids | cut -c1-110 >> $O/conv_layers.txt
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_nets_golden.py -x -q -k "conv or flownet or ffwm" > $O/pytest_conv.txt 2>&1; echo "rc $?" >> $O/pytest_conv.txt
cat $O/conv_layers.txt; tail -n 4 $O/pytest_conv.txt
