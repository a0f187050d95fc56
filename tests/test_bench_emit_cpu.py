"""The bench's output contract: the LAST stdout line is one compact JSON object the driver can parse (round 3's single 23 KB
line was not: BENCH_r03.json parsed = null)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _stub():
    """round 3's full result object (the one the driver failed to parse) as the emitter's input"""
    with open(os.path.join(ROOT, "profiles", "r03_bench_default.json")) as f:
        return json.load(f)


def test_last_line_is_compact_and_has_the_contract_keys_in_order():
    lines = bench.emit_lines(_stub())
    assert len(lines) == 3 and all("\n" not in ln for ln in lines)
    last = lines[-1]
    assert len(last) < 4096, len(last)
    d = json.loads(last)
    assert tuple(d)[:len(CONTRACT)] == CONTRACT
    assert d["config"]["workload"].startswith("BASELINE configs[2]")
    assert "model" not in d["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-3
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert d["roofline_mfma"]["bound"] == "mfma"
    assert set(d["warp_attention_path"]) >= {"fwd_img_per_s", "fwd_bwd_img_per_s"}
    assert abs(d["value"] - d["config"]["global_batch"] / d["ms_per_step"] * 1e3) < 0.5


def test_verbose_lines_keep_everything():
    stub = _stub()
    lines = bench.emit_lines(stub)
    kernels = json.loads(lines[0])["kernels"]
    assert len(kernels) == len(stub["kernels"])
    detail = json.loads(lines[1])["detail"]
    assert "subpaths" in detail and detail["roofline_full"] == stub["roofline"]


def test_oversized_fields_never_push_the_last_line_over_the_limit():
    stub = _stub()
    stub["config"]["launch"] = "x" * 5000
    stub["cpu_baseline"]["sample"] = "y" * 5000
    stub["allreduce"] = {"note": "z" * 3000}
    last = bench.emit_lines(stub)[-1]
    assert len(last) < 4096
    assert tuple(json.loads(last))[:len(CONTRACT)] == CONTRACT


def _r04_rows():
    """round 4's kernel rows and sub-path figures (the verbose lines of profiles/r04_bench_default.json)"""
    with open(os.path.join(ROOT, "profiles", "r04_bench_default.json")) as f:
        lines = [ln for ln in f if ln.strip()]
    return json.loads(lines[0])["kernels"], json.loads(lines[1])["detail"]["subpaths"]


def test_last_line_carries_the_per_config_operator_rows():
    """BASELINE configs[0], [1], [4] must be visible in the line the driver parses."""
    stub = _stub()
    stub["kernels"], stub["subpaths"] = _r04_rows()
    stub["kernel_rows_from"] = "2 eager single-stream steps after the timed region"
    last = bench.emit_lines(stub)[-1]
    assert len(last) < 4096, len(last)
    d = json.loads(last)
    assert tuple(d)[:len(CONTRACT)] == CONTRACT
    ops = d["ops"]
    for k in ("cfg1_rs_fwd", "cfg1_rs_bwd1", "cfg1_rs_bwd2", "cfg5_be_fwd", "cfg5_be_bwd", "cfg5_lar", "warp_fwd@256", "warp_bwd_flow@256",
              "flownet_fwd_cfg2", "rs_bwd1@512", "warp_bwd_feat@256"):
        assert k in ops and len(ops[k]) == 2 and ops[k][0] > 0, (k, ops.get(k))
    # the figures are the rows' own: round 4's cfg-5 extractor forward (211 us = 0.80 by HIP events), backward 596 us = 0.31
    assert abs(ops["cfg5_be_fwd"][0] - 211.05) < 0.01 and abs(ops["cfg5_be_bwd"][1] - 0.3105) < 1e-4
    assert abs(ops["flownet_fwd_cfg2"][0] - 897.2) < 0.1
    assert "after the timed region" in d["roofline"]["rows_from"]


def test_block_attention_backward_row_is_the_sum_of_its_launches():
    """Round 6: cfg5_battn_bwd = far + d(source) + d(flow, weights) launches, priced against the operator's bytes (the first scope's)."""
    where = "cfg5/GPU block attention k=3 backward"
    rows = [{"kernel": "block_attention_bwd_src", "where": where, "avg_us": 260.0, "alg_MB": 416.0, "frac_hbm_peak": 0.2},
            {"kernel": "block_attention_bwd_pix", "where": where, "avg_us": 120.0, "alg_MB": 300.0, "frac_hbm_peak": 0.3},
            {"kernel": "block_attention_bwd_far", "where": where, "avg_us": 8.0, "alg_MB": 2.0, "frac_hbm_peak": 0.01},
            {"kernel": "block_attention_bwd_src", "where": "timed region", "avg_us": 1.0, "alg_MB": 1.0, "frac_hbm_peak": 0.5}]
    ops = bench.ops_summary(rows)
    assert ops["cfg5_battn_bwd"][0] == 388.0
    assert abs(ops["cfg5_battn_bwd"][1] - 416e6 / 388e-6 / bench.HBM_PEAK) < 1e-4
    assert "cfg5_battn_bwd" not in bench.ops_summary(rows[1:])          # without the d(source) launch there is no row


def test_operator_rows_survive_the_size_limit_longer_than_the_optional_objects():
    stub = _stub()
    stub["kernels"], stub["subpaths"] = _r04_rows()
    stub["allreduce"] = {"note": "z" * 3000}
    d = json.loads(bench.emit_lines(stub)[-1])
    assert "ops" in d and "allreduce" not in d
