#!/bin/bash
# round 5, diagnosis batch 2: what corrupts the tiled weight gradient's result in the five-graph ("segments") replay
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/diag2
mkdir -p $O
run() { name=$1; shift; env "$@" FFWM_PROBE_TRACE=1 timeout 600 python tools/dp_capture_probe.py gloo segments 2 2>&1 | grep -v "per-param\|amdgpu.ids\|hostname\|Gloo" | cut -c1-700 > $O/$name.txt; echo "rc $?" >> $O/$name.txt; }
run A_streams FFWM_SEG_STREAMS=1
run B_streams_nocoll FFWM_SEG_STREAMS=1 FFWM_PROBE_NOCOLL=1
run C_streams_arena FFWM_SEG_STREAMS=1 FFWM_PROBE_ARENA=1
run D_streams_ownpools FFWM_SEG_STREAMS=1 FFWM_SEG_OWN_POOLS=1
run E_flow_only FFWM_SEG_STREAMS=flow
run F_loss_only FFWM_SEG_STREAMS=loss
run G_no_streams FFWM_SEG_STREAMS=0
run H_streams_unsliced FFWM_SEG_STREAMS=1 FFWM_OPTS=conv_wgrad_unsliced=1
timeout 1200 python tools/wgrad_stress.py 200 3 3 2 > $O/stress.txt 2>&1; echo "stress rc $?" >> $O/stress.txt
timeout 1500 python tools/ingraph_repeat.py 10 2>&1 | grep "^run\|ingraph_repeat" | cut -c1-900 > $O/ingraph_repeat.txt
grep -c "wrong elements [1-9]" $O/*.txt
tail -n 4 $O/stress.txt $O/ingraph_repeat.txt
