#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/run14
mkdir -p $O
RS_BWD1_FIXED=0 timeout 600 python tools/rs_bwd1_ablate.py 2>&1 | grep -v amdgpu.ids > $O/rs_bwd1_ablate_fixed.txt
RS_BWD1_FIXED=2 timeout 600 python tools/rs_bwd1_ablate.py 2>&1 | grep -v amdgpu.ids > $O/rs_bwd1_ablate_double.txt
cat $O/rs_bwd1_ablate_fixed.txt $O/rs_bwd1_ablate_double.txt
