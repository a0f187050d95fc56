#!/bin/bash
# Round 6, GPU-side half of the profile refresh (run through gpurun from the repo root; ~8 min of box time).  Everything lands in
# gpurun_out/refresh/ (small CSV / JSON / text); tools/fold_profiles.py turns it into profiles/r06_*.
#   1. rocprofv3 --kernel-trace of the default train bench: steady-state per-kernel table (last 200 ms) + per-step accounting
#   2. the operator workloads: `--workload ops` (stand-alone cfg-1 / cfg-5 / HBM-resident shapes, 10 launches each behind a cache flush)
#      and `--workload warp`, aggregated over the WHOLE trace, per kernel AND launch geometry, this library's kernels only, the first
#      two launches of every row dropped (round 5's "last 30 ms" window had caught only the flush copies)
#   3. warpatt / flownet / flowtrain steady-state tables
#   4. counter passes of the default command (--kernel-include-regex ffwm; FETCH_SIZE, WRITE_SIZE, two SQ sets: SEPARATE runs) and the
#      FETCH / WRITE calibration on a known byte count (tools/pmc_calib.py)
#   5. the default and the eager bench lines
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/refresh
rm -rf $OUT && mkdir -p $OUT
trace() {   # tag window-ms extra-steady-args -- bench-args...
  local TAG=$1 WIN=$2 EXTRA=$3; shift; shift; shift
  rm -rf /tmp/kt_$TAG && mkdir -p /tmp/kt_$TAG
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$TAG -- python $R/bench.py "$@" --no-cpu-baseline --no-extras > $OUT/${TAG}_bench.log 2>&1
  local F=$(find /tmp/kt_$TAG -name "*kernel_trace.csv" | head -1)
  python $R/tools/steady_stats.py $F $OUT/${TAG}_kernel_stats.csv --window-ms $WIN $EXTRA \
      --header "rocprofv3 --kernel-trace -- python bench.py $* --no-cpu-baseline --no-extras; steady_stats.py --window-ms $WIN $EXTRA (tools/r06/refresh.sh)" > $OUT/${TAG}_steady.log 2>&1
  grep -h '^{' $OUT/${TAG}_bench.log | tail -1 > $OUT/${TAG}_bench.json
}
trace train_step 200 "" --steps 6 --warmup 3 --no-kernels
python $R/tools/step_trace.py $(find /tmp/kt_train_step -name "*kernel_trace.csv" | head -1) --steps 3 --top 60 > $OUT/train_step_per_step.txt 2>&1
FFWM_BENCH_SKIP_FALLBACK=1 trace ops 0 "--by-shape --include ffwm:: --skip-first 2" --workload ops --steps 10 --warmup 3
trace warp 0 "--by-shape --include ffwm:: --skip-first 2" --workload warp --steps 40 --warmup 10 --no-kernels
trace warpatt 30 "" --workload warpatt --steps 20 --warmup 5 --no-kernels
trace flownet 15 "" --workload flownet --steps 40 --warmup 10 --no-kernels
trace flowtrain 60 "" --workload flowtrain --steps 10 --warmup 3 --no-kernels
# whole-run statistics of the hand-written kernels of the default command in eager mode (train steps + stand-alone shapes)
rm -rf /tmp/kt_all && mkdir -p /tmp/kt_all
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_all -- python $R/bench.py --graph off --steps 6 --warmup 3 --no-cpu-baseline --no-extras > $OUT/trace_all.log 2>&1
python $R/tools/steady_stats.py $(find /tmp/kt_all -name "*kernel_trace.csv" | head -1) $OUT/ffwm_kernels_whole_run.csv --window-ms 0 --by-shape --include "ffwm::" \
    --header "hand-written kernels over the whole traced run, per launch geometry: rocprofv3 --kernel-trace -- python bench.py --graph off --steps 6 --warmup 3 --no-cpu-baseline --no-extras" > /dev/null 2>&1
# counter passes
PMC="python $R/bench.py --graph off --steps 2 --warmup 2 --no-cpu-baseline --no-extras"
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SALU"; do
  rm -rf /tmp/pmc_$i && mkdir -p /tmp/pmc_$i
  timeout 600 rocprofv3 --pmc $C --kernel-include-regex "ffwm" --kernel-trace --output-format csv -d /tmp/pmc_$i -- $PMC > $OUT/pmc_pass$i.log 2>&1 || echo "counter pass $i ($C) failed rc=$?" | tee -a $OUT/pmc_failures.txt
  i=$((i+1))
done
python $R/tools/pmc_fold.py $OUT/bench_pmc_raw.json /tmp/pmc_0 /tmp/pmc_1 /tmp/pmc_2 /tmp/pmc_3 > $OUT/pmc_fold.txt 2>&1
# calibration: a kernel that reads and writes a KNOWN number of bytes (268 MB each way), 16-byte and 4-byte accesses
for j in 0 1; do
  C=$([ $j = 0 ] && echo FETCH_SIZE || echo WRITE_SIZE)
  rm -rf /tmp/pmc_cal_$j && mkdir -p /tmp/pmc_cal_$j
  timeout 300 rocprofv3 --pmc $C --kernel-include-regex "bias_act" --kernel-trace --output-format csv -d /tmp/pmc_cal_$j -- python $R/tools/pmc_calib.py > $OUT/pmc_calib_pass$j.log 2>&1
done
python $R/tools/pmc_fold.py $OUT/pmc_calibration_raw.json /tmp/pmc_cal_0 /tmp/pmc_cal_1 > $OUT/pmc_calib_fold.txt 2>&1
timeout 1200 python $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 300 $OUT/bench_default.json
timeout 900 python $R/bench.py --graph off --no-cpu-baseline --no-kernels --no-extras > $OUT/bench_eager.json 2>/dev/null
timeout 600 python $R/tools/wino_check.py 2>&1 | grep -v amdgpu.ids > $OUT/winograd_vs_vendor.txt
(cd $R && timeout 900 python tools/ref_vs_hip.py > $OUT/ref_vs_hip.log 2>&1; cp gpurun_out/ref_vs_hip.json $OUT/ref_vs_hip.json)
ls -la $OUT | head -50
