#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/run18
mkdir -p $O
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?" >> $O/bench_default.err
tail -n 2 $O/bench_default.err; tail -n 1 $O/bench_default.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(len(json.dumps(d)), d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_mfma']['frac']); print(json.dumps(d['ops']))"
