#!/bin/bash
# rocprofv3 kernel trace of the guided-filter launches (tools/gf_time.py) -> average duration per kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gfkt
GF_NOPROF=1 timeout 180 rocprofv3 --kernel-trace --output-format csv -d /tmp/gfkt -- python $R/tools/gf_time.py > /tmp/gfkt.log 2>&1 || { echo "rocprofv3 failed"; tail -5 /tmp/gfkt.log; }
F=$(find /tmp/gfkt -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "gf_" in r["Kernel_Name"]:
        import re
        d[re.search(r"gf_\w+<[^>]*>", r["Kernel_Name"]).group(0)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v = v[len(v) // 2:]
    print("%-62s n=%3d avg %.2f us" % (k, len(v), sum(v) / len(v)))
PY
