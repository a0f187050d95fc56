#!/bin/bash
# round 5, diagnosis batch 3: the five-graph replay with the library's own zero-fill kernel (default now) against hipMemsetAsync nodes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/diag3
mkdir -p $O
run() { name=$1; shift; env "$@" FFWM_PROBE_TRACE=1 timeout 600 python tools/dp_capture_probe.py gloo segments 4 2>&1 | grep -E "^rank|wrong elements [1-9]|Error|error" | sed -e 's/{.G.*weights/weights/' | cut -c1-300 > $O/$name.txt; echo "rc $?" >> $O/$name.txt; }
for i in 1 2 3; do run fill_kernel_$i FFWM_SEG_STREAMS=1; done
for i in 1 2; do run memset_node_$i FFWM_SEG_STREAMS=1 FFWM_OPTS=zero_fill_memset=1; done
run fill_kernel_notrace FFWM_SEG_STREAMS=1 FFWM_PROBE_TRACE=0
FFWM_SEG_STREAMS=1 timeout 900 python tools/dp_capture_probe.py gloo segments 200 2>&1 | grep -E "^rank" | sed -e 's/{.G.*weights/weights/' | cut -c1-200 | awk 'NR<=3 || NR%20==0' > $O/fill_kernel_200_replays.txt
FFWM_SEG_STREAMS=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 tools/dp_capture_probe.py gloo segments 20 2>&1 | grep -E "^rank" | sed -e 's/{.G.*weights/weights/' | cut -c1-200 | awk 'NR<=4 || NR%8==0' > $O/fill_kernel_2ranks.txt
tail -n 3 $O/*.txt
