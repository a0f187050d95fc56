#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/run5
mkdir -p $O
timeout 600 python tools/be_bwd_pair_ab.py > $O/pair_ab.txt 2>&1; echo "rc $?" >> $O/pair_ab.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_golden.py tests/test_gpu_canary.py -x -q -k "block or attention or live_resample or extractor" > $O/pytest_be.txt 2>&1; echo "rc $?" >> $O/pytest_be.txt
tail -n 25 $O/pair_ab.txt; tail -n 8 $O/pytest_be.txt
