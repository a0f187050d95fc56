#!/bin/bash
# round 5, batch 4: the whole GPU suite on the zero-fill fix + the five-graph replay over 1000 replays (1 rank, 2 gloo ranks)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/run4
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
timeout 900 python tools/dp_capture_probe.py gloo segments 1000 2>&1 | grep -E "^rank" | sed -e 's/{.G.*weights/weights/' | cut -c1-200 | awk 'NR<=3 || NR%100==0' > $O/segments_1000_replays_1rank.txt
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29578 tools/dp_capture_probe.py gloo segments 500 2>&1 | grep -E "^rank" | sed -e 's/{.G.*weights/weights/' | cut -c1-200 | awk 'NR<=4 || NR%100==0' > $O/segments_500_replays_2ranks.txt
timeout 600 python tools/ingraph_repeat.py 6 2>&1 | grep "^run\|ingraph_repeat" | cut -c1-500 > $O/ingraph_repeat.txt
tail -n 6 $O/*.txt
