#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/run11
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?" >> $O/bench_default.err
tail -n 6 $O/pytest_gpu.txt; tail -n 3 $O/bench_default.err; tail -n 1 $O/bench_default.json | cut -c1-3900
