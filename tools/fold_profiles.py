"""Local half of the profile refresh: gpurun_out/refresh/ (tools/refresh_profiles.sh) -> profiles/r01_*.

    python tools/fold_profiles.py [gpurun_out/refresh]

Maps the raw per-kernel PMC averages onto the launch scopes bench.py reports (ffwm_prof_* names), applies
the FETCH_SIZE x2 calibration of this box (profiles/r01_kbench_pmc_raw.json: a 1.2 GB copy reports half its
read bytes), and copies the kernel statistics and the default bench line."""
import json
import os
import shutil
import sys

SCOPES = {   # launch scope -> kernel-name prefix (template arguments included where they select the variant)
    "adam_flat": "adam_flat_kernel",
    "bn_lrelu_bwd": "bn_lrelu_bwd_kernel<1024>",
    "bn_lrelu_fwd": "bn_lrelu_fwd_kernel<1024>",
    "block_extractor_bwd_far": "be_bwd_far2_kernel<float, 3, false>",
    "block_extractor_bwd_tile2": "be_bwd_tile2_kernel<3, 32, 4, false>",
    "block_extractor_fwd_lds": "be_fwd_lds_kernel<float, 3, 4, 0>",
    "conv3x3_wgrad": "conv3x3_wgrad_kernel<false>",
    "conv3x3_wgrad_packed": "conv3x3_wgrad_kernel<true>",
    "guided_filter_bwd": "gf_backward_kernel<float>",
    "guided_filter_fwd": "gf_forward_kernel<float>",
    "mfm_bwd": "mfm_bwd",
    "mfm_fwd": "mfm_fwd",
    "local_attn_reshape_bwd": "lar_bwd_kernel<float, 3, false>",
    "local_attn_reshape_fwd": "lar_fwd_kernel<float, 3>",
    "resample2d_bwd_input1_plane": "rs_bwd1_plane_kernel<float, 2>",
    "resample2d_bwd_input2": "rs_bwd2_kernel<float, 2>",
    "resample2d_fwd": "rs_fwd_kernel<float, 2>",
    "spectral_norm_bwd_apply": "sn_bwd_apply_kernel<float>",
    "spectral_norm_bwd_dot": "sn_bwd_dot_kernel<float>",
    "spectral_norm_fwd_div": "sn_phase3_kernel<float>",
    "spectral_norm_fwd_wtu": "sn_phase1_kernel<float>",
    "spectral_norm_fwd_wv": "sn_phase2_kernel<float>",
    "warp_bwd_feat": "warp_bwd_feat_plane_kernel<float, false",
    "warp_bwd_flow": "warp_bwd_kernel<float, false>",
    "warp_flipcat_bwd_feat": "warp_bwd_feat_plane_kernel<float, true",
    "warp_flipcat_bwd_flow": "warp_bwd_kernel<float, true>",
    "warp_flipcat_fwd": "warp_fwd_kernel<float, true>",
    "warp_fwd": "warp_fwd_kernel<float, false>",
}
FETCH_CORRECTION = 2.0


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/refresh"
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    raw = json.load(open(os.path.join(src, "bench_pmc_raw.json")))
    if not raw:
        print("no PMC data in %s (counter passes failed?): profiles/r01_pmc_traffic.json left as it is" % src)
    out = {
        "_about": "HBM-side traffic per dispatch from rocprofv3 PMC passes of `python bench.py --steps 2 --warmup 2 "
                  "--no-cpu-baseline` (one pass with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE; they do not fit one pass; "
                  "tools/refresh_profiles.sh + tools/fold_profiles.py). Averages over the dispatches of each launch scope (the "
                  "in-step scopes mix layer shapes exactly as bench.py's averages do; flow nets fitted to the identity grid as in "
                  "the default bench). Units KiB as reported. Calibration on this box (profiles/r01_kbench_pmc_raw.json): a 1.2 GB "
                  "copy (1,179,648 KiB read + written) reports FETCH_SIZE 589,824 KiB and WRITE_SIZE 1,179,648 KiB, so "
                  "traffic_bytes = (2 x fetch_KiB + write_KiB) x 1024.",
        "_calibration": {"copy_1.2GB_expected_KiB_each_way": 1179648, "copy_fetch_KiB": 589824.0, "copy_write_KiB": 1179648.0,
                         "fetch_correction": FETCH_CORRECTION},
    }
    for scope, prefix in sorted(SCOPES.items()):
        rows = [(k, v) for k, v in raw.items() if ("::" + prefix) in k]
        if not rows:
            continue
        n = sum(v["dispatches"] for _, v in rows)
        fetch = sum(v["fetch_KiB"] * v["dispatches"] for _, v in rows) / n
        write = sum(v["write_KiB"] * v["dispatches"] for _, v in rows) / n
        out[scope] = {"kernel": prefix, "dispatches": n, "fetch_KiB": round(fetch, 1), "write_KiB": round(write, 1),
                      "traffic_bytes": int((FETCH_CORRECTION * fetch + write) * 1024)}
    if raw:
        json.dump(out, open(os.path.join(root, "profiles", "r01_pmc_traffic.json"), "w"), indent=1, sort_keys=True)
    for a, b in (("train_step_kernel_stats.csv", "r01_train_step_kernel_stats.csv"),
                 ("ffwm_kernels_whole_run.csv", "r01_ffwm_kernels_rocprofv3.csv"),
                 ("bench_default.json", "r01_bench_default.json")):
        shutil.copy(os.path.join(src, a), os.path.join(root, "profiles", b))
    for k, v in out.items():
        if not k.startswith("_"):
            print("%-30s %10.3f MB  x%d" % (k, v["traffic_bytes"] / 1e6, v["dispatches"]))


if __name__ == "__main__":
    main()
