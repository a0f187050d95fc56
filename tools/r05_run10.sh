#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/run10
mkdir -p $O
timeout 600 python tools/rs_bwd1_fixed_ab.py 2>&1 | grep -v amdgpu.ids > $O/rs_fixed_ab.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_golden.py -x -q -k "resample" > $O/pytest_rs.txt 2>&1; echo "rc $?" >> $O/pytest_rs.txt
cat $O/rs_fixed_ab.txt; tail -n 12 $O/pytest_rs.txt
