cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_SMEM"; do
  rm -rf /tmp/p_$i; mkdir -p /tmp/p_$i
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/p_$i -- python $R/tools/wino_one.py > /tmp/p_$i.log 2>&1 || { echo "pass $i failed"; tail -3 /tmp/p_$i.log; }
  i=$((i+1))
done
python $R/tools/pmc_fold.py /tmp/conv1.json /tmp/p_0 /tmp/p_1 /tmp/p_2 /tmp/p_3 | grep -A1 winograd_conv_raw
