#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/run7
mkdir -p $O
timeout 600 python tools/be_bwd_fixed_ab.py 2>&1 | grep -v amdgpu.ids > $O/fixed_ab.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_golden.py tests/test_gpu_canary.py -x -q -k "block or attention or extractor" > $O/pytest_be.txt 2>&1; echo "rc $?" >> $O/pytest_be.txt
cat $O/fixed_ab.txt; tail -n 12 $O/pytest_be.txt
