#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/run8
mkdir -p $O
BE_BWD_FIXED=0 timeout 600 python tools/be_bwd_ablate.py 2>&1 | grep -v amdgpu.ids > $O/ablate_fixed.txt
BE_BWD_FIXED=2 timeout 600 python tools/be_bwd_ablate.py 2>&1 | grep -v amdgpu.ids > $O/ablate_double.txt
cat $O/ablate_fixed.txt $O/ablate_double.txt
