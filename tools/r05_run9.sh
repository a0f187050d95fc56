#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/run9
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "fixed_point" > $O/pytest_fixed.txt 2>&1; echo "rc $?" >> $O/pytest_fixed.txt
tail -n 30 $O/pytest_fixed.txt
