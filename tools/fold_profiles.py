"""Local half of the profile refresh: gpurun_out/refresh/ (tools/r06/refresh.sh; rounds 2-5: tools/refresh_profiles.sh) -> profiles/r06_*.

    python tools/fold_profiles.py [gpurun_out/refresh]

Maps the raw per-kernel PMC averages (tools/pmc_fold.py) onto the launch scopes bench.py reports (ffwm_prof_* names),
applies the FETCH_SIZE x2 calibration of this box (profiles/r01_kbench_pmc_raw.json: a 1.2 GB copy reports half its read
bytes), and copies the kernel statistics and the default bench line.

Refuses to publish what was not measured: the traffic file is only written when BOTH counter passes produced rows, and a
scope whose FETCH_SIZE or WRITE_SIZE average is missing or zero is left out (round 1 published write-only byte counts as
"traffic" after the FETCH pass had aborted)."""
import json
import os
import shutil
import sys

SCOPES = {   # launch scope -> kernel-name fragment (template arguments included where they select the variant)
    "adam_flat": "adam_flat_kernel",
    "bn_lrelu_bwd": "bn_lrelu_bwd_kernel<1024>",
    "bn_lrelu_fwd": "bn_lrelu_fwd_kernel<1024>",
    "block_extractor_bwd_far": "be_bwd_far2_kernel<float, 3, false>",
    "block_extractor_bwd_tile2": "be_bwd_tile2_kernel<3, 32, 4, false,",
    "block_extractor_fwd_lds": "be_fwd_lds_kernel<float, 3, 4, 0>",
    "conv3x3_thin_tail": "conv3x3_thin_kernel<3>",
    "conv3x3_wgrad": "conv3x3_wgrad_kernel<false>",
    "conv3x3_wgrad_winograd": "conv3x3_wgrad_wino_kernel",
    "conv3x3_wgrad_packed": "conv3x3_wgrad_kernel<true>",
    "conv_fwd_mfma": "conv_fwd_kernel<0,",
    "conv_fwd_mfma_transposed": "conv_fwd_kernel<1,",
    "conv_winograd_dgrad": "winograd_conv_raw_kernel<0>",
    "conv_winograd_fwd": "winograd_conv_raw_kernel<0>",
    "conv_winograd_weights": "winograd_weights_kernel",
    # four launches per scope: the traffic of a scope is the SUM over its kernels
    "guided_filter_bwd": ("gf_cols_kernel<float, 2>", "gf_rows_kernel<float, 2>", "gf_cols_kernel<float, 3>", "gf_rows_kernel<float, 3>"),
    "guided_filter_fwd": ("gf_cols_kernel<float, 0>", "gf_rows_kernel<float, 0>", "gf_cols_kernel<float, 1>", "gf_rows_kernel<float, 1>"),
    "mfm_bwd": "mfm_bwd4_kernel",
    "mfm_fwd": "mfm_fwd4_kernel",
    "local_attn_reshape_bwd": "lar_bwd_kernel<float, 3, false>",
    "local_attn_reshape_fwd": "lar_fwd_kernel<float, 3>",
    "resample2d_bwd_input1_plane": "rs_bwd1_plane_kernel<float, 2>",
    "resample2d_bwd_input1_owned": ("rs_bwd1_owned_kernel<2>", "rs_bwd1_far_kernel<2>"),        # round 6: one scope over both launches
    "conv_fwd_split_reduce": "conv_split_reduce_kernel<4>",
    "resample2d_bwd_input2": "rs_bwd2_kernel<float, 2>",
    "resample2d_bwd_input2_lds": "rs_bwd2_lds_kernel<2,",
    "resample2d_fwd_lds": "rs_fwd_lds_kernel<2,",
    "spectral_norm_bwd_apply": "sn_bwd_apply_kernel<float>",
    "spectral_norm_bwd_dot": "sn_bwd_dot_kernel<float>",
    "spectral_norm_fwd_div": "sn_phase3_kernel<float>",
    "spectral_norm_fwd_wtu": "sn_phase1_kernel<float>",
    "spectral_norm_fwd_wv": "sn_phase2_kernel<float>",
    "warp_flipcat_fwd_multi": "warp_fwd_multi_kernel<float, true>",
    "warp_flipcat_bwd_flow@256": "warp_bwd_flow_lds_kernel<true>",
    "warp_fwd_multi": "warp_fwd_multi_kernel<float, false>",
    "warp_flipcat_bwd_flow_multi": "warp_bwd_flow_multi_kernel<float, true>",
    "warp_bwd_flow_multi": "warp_bwd_flow_multi_kernel<float, false>",
    "warp_flipcat_bwd_feat@128": "warp_bwd_feat_plane_kernel<float, true, 1>",
    "warp_flipcat_bwd_feat@64": "warp_bwd_feat_plane_kernel<float, true, 2>",
    "warp_flipcat_bwd_feat@32": "warp_bwd_feat_plane_kernel<float, true, 8>",
    # round 3
    "conv_wgrad_mfma_tiled": "conv_wgrad_tile_kernel<",
    "l1_multi_fwd": "l1_multi_kernel<false>",
    "l1_multi_bwd": "l1_multi_kernel<true>",
    "flownet_flow_head": "flow_head_kernel<",
    "flownet_flow_head_bwd": "flow_head_bwd_kernel<",
    "flownet_flow_up": "flow_up_kernel",
    "flownet_flow_up_bwd": "flow_up_bwd_kernel",
    "resample2d_bwd_input1_tile": "rs_bwd1_tile_kernel<2,",
    # (large ks-4 calls launch BOTH d_input1 kernels and one returns at once: these two per-dispatch averages include such dispatches)
    "resample2d_bwd_input1_taplane": "rs_bwd1_taplane_kernel<2,",
    "warp_flipcat_fwd@256": "warp_fwd_lds_kernel<true>",
    "warp_flipcat_bwd_feat_tile@256": "warp_bwd_feat_tile_kernel<true, 2,",
    "warp_flipcat_bwd_feat_far@256": "warp_bwd_feat_far_kernel<true>",
    "warp_flipcat_bwd_flow@256": "warp_bwd_kernel<float, true>",
    "block_attention_fwd_lds": "ba_fwd_pix_kernel<",
    # round 6: the block attention backward by linearity
    "block_attention_bwd_src": "ba_bwd_src_kernel<",
    "block_attention_bwd_pix": "ba_bwd_pix_kernel<",
}
FETCH_CORRECTION = 2.0
ROUND = "r06"


def calibration(src):
    """FETCH_SIZE / WRITE_SIZE of THIS refresh's calibration passes (tools/pmc_calib.py: bias_act_kernel reads and writes 262,144 KiB with
    16-byte accesses, 262,080 KiB with 4-byte ones) beside the byte counts they must show: the measured ratio is what justifies
    `fetch_correction` (round 5 left this object empty)."""
    cal = {"fetch_correction": FETCH_CORRECTION, "expected_KiB_each_way": {"bias_act_kernel<0, 4>": 262144.0, "bias_act_kernel<0, 1>": 262080.0}}
    path = os.path.join(src, "pmc_calibration_raw.json")
    if not os.path.exists(path):
        cal["measured"] = "no calibration pass in this refresh"
        return cal
    raw = json.load(open(path))
    meas = {}
    for k, v in raw.items():
        for frag in cal["expected_KiB_each_way"]:
            if frag in k:
                exp = cal["expected_KiB_each_way"][frag]
                meas[frag] = {"FETCH_SIZE_KiB": round(v.get("FETCH_SIZE", 0.0), 1), "WRITE_SIZE_KiB": round(v.get("WRITE_SIZE", 0.0), 1),
                              "dispatches": v.get("dispatches"),
                              "fetch_reported_over_read": round(v.get("FETCH_SIZE", 0.0) / exp, 4) if exp else None,
                              "write_reported_over_written": round(v.get("WRITE_SIZE", 0.0) / exp, 4) if exp else None}
    cal["measured"] = meas or "calibration kernels not found in the counter output"
    return cal


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/refresh"
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    prof = os.path.join(root, "profiles")
    raw_path = os.path.join(src, "bench_pmc_raw.json")
    raw = json.load(open(raw_path)) if os.path.exists(raw_path) else {}
    have_fetch = any(v.get("FETCH_SIZE", 0) > 0 for v in raw.values())
    have_write = any(v.get("WRITE_SIZE", 0) > 0 for v in raw.values())
    out = {
        "_about": "HBM-side traffic per dispatch from rocprofv3 counter passes of `python bench.py --steps 2 --warmup 2 "
                  "--no-cpu-baseline --no-extras` (--pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs -- they do not fit one "
                  "pass -- with --kernel-include-regex ffwm; tools/refresh_profiles.sh + tools/fold_profiles.py).  Averages over "
                  "the dispatches of each launch scope (in-step scopes: flow nets fitted to the identity grid as in the default "
                  "bench; multi-problem warp launches cover all their levels).  Units KiB as reported.  Calibration on this box "
                  "(profiles/r01_kbench_pmc_raw.json): a 1.2 GB copy (1,179,648 KiB read + written) reports FETCH_SIZE 589,824 "
                  "KiB and WRITE_SIZE 1,179,648 KiB, so traffic_bytes = (2 x fetch_KiB + write_KiB) x 1024.  Scopes without a "
                  "non-zero FETCH_SIZE and WRITE_SIZE measurement are omitted.",
        "_calibration": calibration(src),
    }
    dropped = []
    for scope, frag in sorted(SCOPES.items()):
        if isinstance(frag, tuple):              # a scope of several launches: sum of the kernels' per-dispatch averages
            parts = []
            for f in frag:
                rs = [v for k, v in raw.items() if ("::" + f) in k]
                if rs:
                    m = sum(v["dispatches"] for v in rs)
                    parts.append((m, sum(v.get("FETCH_SIZE", 0.0) * v["dispatches"] for v in rs) / m,
                                  sum(v.get("WRITE_SIZE", 0.0) * v["dispatches"] for v in rs) / m))
            if len(parts) != len(frag):
                continue
            n = min(p[0] for p in parts)
            fetch, write = sum(p[1] for p in parts), sum(p[2] for p in parts)
            rows = []
            frag = " + ".join(frag)
        else:
            rows = [v for k, v in raw.items() if ("::" + frag) in k]
            if not rows:
                continue
            n = sum(v["dispatches"] for v in rows)
            fetch = sum(v.get("FETCH_SIZE", 0.0) * v["dispatches"] for v in rows) / n
            write = sum(v.get("WRITE_SIZE", 0.0) * v["dispatches"] for v in rows) / n
        if fetch <= 0 or write <= 0:
            dropped.append(scope)
            continue
        row = {"kernel": frag, "dispatches": n, "fetch_KiB": round(fetch, 1), "write_KiB": round(write, 1),
               "traffic_bytes": int((FETCH_CORRECTION * fetch + write) * 1024)}
        for c in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                  "SQ_INSTS_VALU", "SQ_INSTS_LDS"):
            vals = [v[c] * v["dispatches"] for v in rows if c in v]
            if vals:
                row[c] = round(sum(vals) / n, 1)
        out[scope] = row
    dst = os.path.join(prof, ROUND + "_pmc_traffic.json")
    if have_fetch and have_write:
        json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
        print("wrote", dst, "(%d scopes; omitted, no valid fetch/write: %s)" % (len(out) - 2, dropped or "none"))
    else:
        print("REFUSED to write %s: FETCH pass %s, WRITE pass %s -- the committed file is left as it is"
              % (dst, "ok" if have_fetch else "EMPTY", "ok" if have_write else "EMPTY"))
    if raw:
        json.dump(raw, open(os.path.join(prof, ROUND + "_pmc_raw_per_kernel.json"), "w"), indent=1, sort_keys=True)
    copies = [("train_step_kernel_stats.csv", "_train_step_kernel_stats.csv"), ("warpatt_kernel_stats.csv", "_warpatt_kernel_stats.csv"),
              ("flownet_kernel_stats.csv", "_flownet_lean_kernel_stats.csv"), ("flownet_module_kernel_stats.csv", "_flownet_module_kernel_stats.csv"),
              ("flowtrain_kernel_stats.csv", "_flowtrain_kernel_stats.csv"), ("ops_kernel_stats.csv", "_ops_kernel_stats.csv"),
              ("ffwm_kernels_whole_run.csv", "_ffwm_kernels_rocprofv3.csv"), ("bench_default.json", "_bench_default.json"),
              ("warpatt_bench.json", "_bench_warpatt.json"), ("warp_kernel_stats.csv", "_warp_kernel_stats.csv"), ("warp_bench.json", "_bench_warp.json"), ("flownet_bench.json", "_bench_flownet.json"),
              ("flowtrain_bench.json", "_bench_flowtrain.json"), ("ops_bench.json", "_bench_ops.json"),
              ("winograd_sq_192.txt", "_winograd_sq_counters_192.txt"), ("winograd_sq_256.txt", "_winograd_sq_counters_256.txt"),
              ("winograd_mem_192.txt", "_winograd_mem_counters_192.txt"), ("winograd_mem_256.txt", "_winograd_mem_counters_256.txt"),
              ("winograd_vs_vendor.txt", "_winograd_vs_vendor.txt"), ("pmc_calibration_raw.json", "_pmc_calibration_raw.json"), ("winograd_layers_of_the_step.txt", "_winograd_layers_of_the_step.txt"),
              ("train_step_per_step.txt", "_train_step_per_step.txt"), ("train_step_eager_per_step.txt", "_train_step_eager_per_step.txt"),
              ("bwd_layers.txt", "_bwd_layers.txt"), ("warp_step_sweep.txt", "_warp_step_sweep.txt"), ("ab_results.txt", "_ab_switches.txt"),
              ("rs_bwd1.txt", "_resample2d_bwd_input1_variants.txt"), ("host_probe.txt", "_host_issue_vs_drain.txt"),
              ("two_stream_check.txt", "_multi_stream_check.txt"), ("bench_eager.json", "_bench_eager.json"),
              ("gf_trace.txt", "_guided_filter_kernels.txt"), ("ref_vs_hip.json", "_ref_vs_hip.json"),
              ("warp_multi_lds_sweep.txt", "_warp_multi_lds_sweep_refresh.txt"), ("warp_bwd_flow_variants.txt", "_warp_bwd_flow_variants.txt"),
              ("be_bwd_ablate.txt", "_be_bwd_ablate_refresh.txt"), ("dp_capture_probe.txt", "_dp_capture_probe.txt")]
    for a, b in copies:
        p = os.path.join(src, a)
        if os.path.exists(p) and os.path.getsize(p) > 0:
            shutil.copy(p, os.path.join(prof, ROUND + b))
        else:
            print("missing:", a)
    for k, v in out.items():
        if not k.startswith("_"):
            print("%-32s %10.3f MB  x%d" % (k, v["traffic_bytes"] / 1e6, v["dispatches"]))


if __name__ == "__main__":
    main()
