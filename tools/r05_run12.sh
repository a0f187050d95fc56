#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/run12
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_nets_golden.py tests/test_gpu_parity.py tests/test_gpu_ref_golden.py -x -q > $O/pytest_gpu_rest.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu_rest.txt
tail -n 8 $O/pytest_gpu_rest.txt
