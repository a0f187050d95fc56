#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/run15
mkdir -p $O
timeout 600 python tools/thin_tail_time.py 2>&1 | grep -v amdgpu.ids > $O/thin_tail.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_nets_golden.py -x -q -k "winograd or thin or ffwm_generator" > $O/pytest_wino.txt 2>&1; echo "rc $?" >> $O/pytest_wino.txt
cat $O/thin_tail.txt; tail -n 4 $O/pytest_wino.txt
