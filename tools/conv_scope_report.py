"""bench.py train step with the routed conv backward on / off: per-scope totals of the hand-written conv kernels."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for dg, wg in (("0", "0"), ("1", "1")):
    env = dict(os.environ, FFWM_CONV_DGRAD=dg, FFWM_CONV_WGRAD=wg)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "8", "--warmup", "3", "--no-cpu-baseline", "--no-kernels", "--no-extras"],
                         env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    print("dgrad=%s wgrad=%s: %.2f img/s, %.2f ms/step" % (dg, wg, d["value"], d["ms_per_step"]))
    for r in d["kernels"]:
        if r["where"] == "timed region" and r["kernel"].startswith("conv"):
            print("   %-28s x%4d/step avg %7.1f us  %6.2f ms/step  %s" % (r["kernel"], r["launches"] / 8, r["avg_us"], r["total_ms"] / 8,
                                                                        ("%.1f TF" % r["TFLOPs"]) if "TFLOPs" in r else ""))
