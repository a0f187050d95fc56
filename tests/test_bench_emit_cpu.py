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
