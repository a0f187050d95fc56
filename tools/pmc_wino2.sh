cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  rm -rf /tmp/q_$i; mkdir -p /tmp/q_$i
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/q_$i -- python $R/tools/wino_one.py > /tmp/q_$i.log 2>&1 || { echo "pass $i failed"; tail -3 /tmp/q_$i.log; }
  i=$((i+1))
done
python $R/tools/pmc_fold.py /tmp/wino2.json /tmp/q_0 /tmp/q_1 /tmp/q_2 /tmp/q_3 /tmp/q_4 | grep -A1 winograd_conv_raw
