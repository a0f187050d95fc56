#!/bin/bash
# round 5, diagnosis batch 1: the tiled weight-gradient kernel under multi-stream graph replay
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/diag1
mkdir -p $O
timeout 1200 python tools/wgrad_stress.py 200 3 3 2 > $O/stress.txt 2>&1; echo "stress rc $?" >> $O/stress.txt
FFWM_SEG_STREAMS=1 timeout 600 python tools/dp_capture_probe.py gloo segments 3 > $O/seg_streams_plain.txt 2>&1; echo "rc $?" >> $O/seg_streams_plain.txt
FFWM_SEG_STREAMS=1 FFWM_PROBE_TRACE=1 FFWM_PROBE_LIST=1 timeout 600 python tools/dp_capture_probe.py gloo segments 3 > $O/seg_streams_trace.txt 2>&1; echo "rc $?" >> $O/seg_streams_trace.txt
FFWM_SEG_STREAMS=1 FFWM_PROBE_TRACE=1 FFWM_PROBE_LIST=1 FFWM_OPTS=conv_wgrad_unsliced=1 timeout 600 python tools/dp_capture_probe.py gloo segments 3 > $O/seg_streams_trace_unsliced.txt 2>&1; echo "rc $?" >> $O/seg_streams_trace_unsliced.txt
timeout 1500 python tools/ingraph_repeat.py 12 > $O/ingraph_repeat.txt 2>&1; echo "rc $?" >> $O/ingraph_repeat.txt
tail -n 5 $O/*.txt
