cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pf; mkdir -p /tmp/pf $R/gpurun_out/pmc
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "ffwm" --kernel-trace --output-format csv -d /tmp/pf -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-kernels > $R/gpurun_out/pmc/try_fetch.log 2>&1; echo rc=$?
tail -3 $R/gpurun_out/pmc/try_fetch.log | cut -c1-300
python $R/tools/pmc_fold.py $R/gpurun_out/pmc/try_fetch.json /tmp/pf | head -30
