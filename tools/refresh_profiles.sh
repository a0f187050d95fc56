#!/bin/bash
# GPU-side half of the profile refresh (run through gpurun from the repo root; ~6 min of box time):
#   1. rocprofv3 --kernel-trace of the default train bench -> steady-state per-kernel stats (last ~2 steps)
#   2. the same for the other workloads (warpatt, flownet, flowtrain, ops)
#   3. counter passes of the default command, hand-written kernels only (--kernel-include-regex ffwm: a counter pass
#      that also instruments MIOpen's kernels aborts with HSA_STATUS_ERROR_INVALID_PACKET_FORMAT): FETCH_SIZE, WRITE_SIZE
#      and two SQ sets in SEPARATE runs (counters + --kernel-trace only)
#   4. the default `python bench.py` line
# Everything lands in gpurun_out/refresh/ (small CSV / JSON only); tools/fold_profiles.py turns it into profiles/r05_*.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/refresh
rm -rf $OUT && mkdir -p $OUT
trace() {   # tag window-ms bench-args...
  local TAG=$1 WIN=$2; shift; shift
  rm -rf /tmp/kt_$TAG && mkdir -p /tmp/kt_$TAG
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$TAG -- python $R/bench.py "$@" --no-cpu-baseline --no-kernels --no-extras > $OUT/${TAG}_bench.log 2>&1
  local F=$(find /tmp/kt_$TAG -name "*kernel_trace.csv" | head -1)
  python $R/tools/steady_stats.py $F $OUT/${TAG}_kernel_stats.csv --window-ms $WIN \
      --header "rocprofv3 --kernel-trace -- python bench.py $* --no-cpu-baseline --no-kernels --no-extras; last $WIN ms of the trace (tools/refresh_profiles.sh)" > $OUT/${TAG}_steady.log 2>&1
  grep -h '^{' $OUT/${TAG}_bench.log | tail -1 > $OUT/${TAG}_bench.json
}
trace train_step 200 --steps 6 --warmup 3
# per-STEP accounting of the same trace (the default = captured step, several streams) and of an eager, single-graph-less run
python $R/tools/step_trace.py $(find /tmp/kt_train_step -name "*kernel_trace.csv" | head -1) --steps 3 --top 60 > $OUT/train_step_per_step.txt 2>&1
rm -rf /tmp/kt_eager && mkdir -p /tmp/kt_eager
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_eager -- python $R/bench.py --graph off --steps 6 --warmup 3 --no-cpu-baseline --no-kernels --no-extras > $OUT/train_eager_bench.log 2>&1
python $R/tools/step_trace.py $(find /tmp/kt_eager -name "*kernel_trace.csv" | head -1) --steps 3 --top 60 > $OUT/train_step_eager_per_step.txt 2>&1
trace warpatt 30 --workload warpatt --steps 20 --warmup 5
trace warp 10 --workload warp --steps 40 --warmup 10
trace flownet 15 --workload flownet --steps 40 --warmup 10
trace flownet_module 20 --workload flownet --flownet-path module --steps 40 --warmup 10
trace flowtrain 60 --workload flowtrain --steps 10 --warmup 3
trace ops 30 --workload ops --steps 10 --warmup 3
# whole-run statistics of the hand-written kernels of the default command (train steps + stand-alone cfg-1 / cfg-5 shapes)
rm -rf /tmp/kt_all && mkdir -p /tmp/kt_all
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_all -- python $R/bench.py --graph off --steps 6 --warmup 3 --no-cpu-baseline --no-extras > $OUT/trace_all.log 2>&1
F=$(find /tmp/kt_all -name "*kernel_trace.csv" | head -1)
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
    f.write("# hand-written kernels over the whole traced run (train steps + stand-alone cfg-1 / cfg-5 shapes), rocprofv3 --kernel-trace -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras\n")
    f.write("Name,Calls,TotalDurationNs,AverageNs\n")
    for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        f.write('"%s",%d,%d,%d\n' % (n, c, t, t // c))
PY
# counter passes
PMC="python $R/bench.py --graph off --steps 2 --warmup 2 --no-cpu-baseline --no-extras"
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SALU"; do
  rm -rf /tmp/pmc_$i && mkdir -p /tmp/pmc_$i
  timeout 600 rocprofv3 --pmc $C --kernel-include-regex "ffwm" --kernel-trace --output-format csv -d /tmp/pmc_$i -- $PMC > $OUT/pmc_pass$i.log 2>&1 || echo "counter pass $i ($C) failed rc=$?" | tee -a $OUT/pmc_failures.txt
  i=$((i+1))
done
python $R/tools/pmc_fold.py $OUT/bench_pmc_raw.json /tmp/pmc_0 /tmp/pmc_1 /tmp/pmc_2 /tmp/pmc_3 > $OUT/pmc_fold.txt 2>&1
timeout 1200 python $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 400 $OUT/bench_default.json
# SQ / memory counters of the Winograd convolution kernel alone (netG's 192 -> 192 and 256 -> 256 layers at 128 x 128, batch 8)
for CK in 192 256; do
  CV_C=$CK CV_K=$CK bash $R/tools/pmc_wino.sh > $OUT/winograd_sq_$CK.txt 2>&1
  CV_C=$CK CV_K=$CK bash $R/tools/pmc_wino2.sh > $OUT/winograd_mem_$CK.txt 2>&1
done
python $R/tools/wino_check.py > $OUT/winograd_vs_vendor.txt 2>&1
python $R/tools/wino_layers.py > $OUT/winograd_layers_of_the_step.txt 2>/dev/null

# round 3: the eager line beside the default (captured) one, every layer's backward vs the vendor's, the in-step warp sweep, switches A/B,
# host issue time vs drain, the multi-stream check
timeout 900 python $R/bench.py --graph off --no-cpu-baseline --no-kernels --no-extras > $OUT/bench_eager.json 2>/dev/null
timeout 900 python $R/tools/bwd_layers.py > $OUT/bwd_layers.txt 2>&1
timeout 600 python $R/tools/warp_step_sweep.py 2>&1 | grep -v Warn > $OUT/warp_step_sweep.txt
timeout 600 python $R/tools/rs_bwd1_cfg1.py 2>&1 | grep -v Warn > $OUT/rs_bwd1.txt
timeout 600 python $R/tools/host_sync_probe.py 2>&1 | grep -E "^host|hip" > $OUT/host_probe.txt
timeout 900 python $R/tools/two_stream_check.py 2>&1 | grep -v Warn | tail -13 > $OUT/two_stream_check.txt
bash $R/tools/gf_trace.sh > $OUT/gf_trace.txt 2>&1
(cd $R && timeout 900 python tools/ref_vs_hip.py > $OUT/ref_vs_hip.log 2>&1; cp gpurun_out/ref_vs_hip.json $OUT/ref_vs_hip.json)
rm -f $R/gpurun_out/ab/results.txt
bash $R/tools/ab_graph.sh "default=X=1" "one_stream=FFWM_STREAMS=0" "five_streams_one_per_branch=FFWM_STREAM_LAYOUT=5" "d_step_on_the_main_stream=FFWM_D_STREAM=0" "fused_residual_off=FFWM_FUSED_RESIDUAL=0" "tiled_wgrad_off=FFWM_TILED_WGRAD=0" "convT_dgrad_off=FFWM_CONVT_DGRAD=0" "conv_dgrad_all_own=FFWM_CONV_DGRAD=1" "bn_fused_from_0=FFWM_BN_MIN_NUMEL=0" "winograd_pairs_100=FFWM_WINOGRAD_MIN_PAIRS=100" > /dev/null 2>&1
cp $R/gpurun_out/ab/results.txt $OUT/ab_results.txt
# round 4: the multi-problem warp launches on direct gathers vs LDS-staged tiles, d(flow) variants per level, block_extractor backward
# with parts switched off, the data-parallel capture modes on the one-GPU box (one-rank RCCL group, forced collectives)
timeout 600 python $R/tools/warp_multi_lds_sweep.py 2>&1 | grep -v "Warn\|amdgpu" > $OUT/warp_multi_lds_sweep.txt
timeout 600 python $R/tools/warp_bwd_flow_variants.py 2>&1 | grep "^(" > $OUT/warp_bwd_flow_variants.txt
timeout 600 python $R/tools/be_bwd_ablate.py 2>&1 | grep "^ablate" > $OUT/be_bwd_ablate.txt
(for m in "nccl probe" "nccl ingraph" "gloo serial"; do echo "=== $m"; timeout 300 python $R/tools/dp_capture_probe.py $m 4 2>&1 | grep -v "amdgpu.ids\|^\[W\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6 | cut -c1-300; done) > $OUT/dp_capture_probe.txt 2>&1
