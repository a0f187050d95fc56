#!/bin/bash
# GPU-side half of the profile refresh (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace of the default train bench -> steady-state per-kernel stats (last 2 steps)
#      and the whole-run stats of the hand-written kernels (incl. the stand-alone cfg-1 / cfg-5 shapes)
#   2. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of the same command (separate passes, counters only)
#   3. the default `python bench.py` line
# Everything lands in gpurun_out/refresh/ (small CSV / JSON only); tools/fold_profiles.py turns it into profiles/.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/refresh
rm -rf /tmp/kt /tmp/pf /tmp/pw && mkdir -p /tmp/kt /tmp/pf /tmp/pw $OUT
CMD="python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline"
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- $CMD --no-kernels > $OUT/trace_bench.log 2>&1
F=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python $R/tools/steady_stats.py $F $OUT/train_step_kernel_stats.csv --window-ms 200 \
    --header "rocprofv3 --kernel-trace -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernels; last 200 ms of the trace = ~2 eager train steps under the tracer (tools/refresh_profiles.sh)" > $OUT/steady.log 2>&1
rm -rf /tmp/kt && mkdir -p /tmp/kt
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- $CMD > $OUT/trace_bench_kernels.log 2>&1
F=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python - "$F" "$OUT/ffwm_kernels_whole_run.csv" <<'PY'
import collections, csv, sys
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "ffwm::" not in n:
        continue
    a = acc.setdefault(n, [0, 0])
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
with open(sys.argv[2], "w") as f:
    f.write("# hand-written kernels over the whole traced run (train steps + stand-alone cfg-1 / cfg-5 shapes), rocprofv3 --kernel-trace\n")
    f.write("Name,Calls,TotalDurationNs,AverageNs\n")
    for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        f.write('"%s",%d,%d,%d\n' % (n, c, t, t // c))
PY
if [ "${FFWM_REFRESH_PMC:-1}" = "1" ]; then
# counter passes: MIOpen's heuristic solvers (FFWM_MIOPEN_DB=0) -- with the tuned solver set a counter pass aborts with
# HSA_STATUS_ERROR_INVALID_PACKET_FORMAT inside a vendor kernel; the traffic of the hand-written kernels does not
# depend on which vendor kernels run next to them.  Short timeouts: a pass takes ~40 s when it works.
PMC="python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline"
FFWM_MIOPEN_DB=0 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -- $PMC > $OUT/pmc_fetch.log 2>&1
FFWM_MIOPEN_DB=0 timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -- $PMC > $OUT/pmc_write.log 2>&1
fi
python $R/tools/pmc_traffic.py /tmp/pf /tmp/pw $OUT/bench_pmc_raw.json > $OUT/pmc_top.txt 2>&1
timeout 900 python $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 300 $OUT/bench_default.json
