#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/run19
mkdir -p $O
FFWM_TEST_POISON=1 timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu_poison.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu_poison.txt
tail -n 12 $O/pytest_gpu_poison.txt
