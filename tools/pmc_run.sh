#!/bin/bash
# Counter passes for ONE command made only of hand-written kernels (no MIOpen: counter passes abort inside vendor kernels).
#   tools/pmc_run.sh <tag> <python command ...>      (run through gpurun from the repo root)
# Separate rocprofv3 runs (counters only + --kernel-trace, never combined with other trace domains):
#   FETCH_SIZE | WRITE_SIZE | SQ set A | SQ set B, folded by tools/pmc_fold.py into gpurun_out/pmc/<tag>.json
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SALU" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  D=/tmp/pmc_${TAG}_$i; rm -rf $D; mkdir -p $D
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- "$@" > $OUT/${TAG}_pass$i.log 2>&1 || echo "pass $i ($C) failed rc=$?"
  i=$((i+1))
done
python $R/tools/pmc_fold.py $OUT/$TAG.json /tmp/pmc_${TAG}_*
