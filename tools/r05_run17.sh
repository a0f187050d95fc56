#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/run17
mkdir -p $O
timeout 900 python tools/wgrad_wino_check.py 2>&1 | grep -v amdgpu.ids > $O/wgrad_wino_check.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_nets_golden.py -x -q -k "wgrad or weight_gradient or routed_gradients" > $O/pytest_wgrad.txt 2>&1; echo "rc $?" >> $O/pytest_wgrad.txt
cat $O/wgrad_wino_check.txt | cut -c1-400; tail -n 4 $O/pytest_wgrad.txt
