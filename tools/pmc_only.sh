cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/refresh
rm -rf /tmp/pf /tmp/pw && mkdir -p /tmp/pf /tmp/pw $OUT
PMC="python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline"
date
FFWM_MIOPEN_DB=0 timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -- $PMC > $OUT/pmc_fetch.log 2>&1; echo rc=$?
date
if grep -q "INVALID_PACKET" $OUT/pmc_fetch.log; then echo "fetch pass crashed"; tail -5 $OUT/pmc_fetch.log; exit 0; fi
FFWM_MIOPEN_DB=0 timeout 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -- $PMC > $OUT/pmc_write.log 2>&1; echo rc=$?
date
python $R/tools/pmc_traffic.py /tmp/pf /tmp/pw $OUT/bench_pmc_raw.json | head -12
