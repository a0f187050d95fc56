"""Backward kernels of every convolution of the FFWM train step (tools/step_conv_calls.txt = the output of tools/step_conv_calls.py),
timed one by one on the GPU: the vendor library's weight gradient / data gradient (each call with the layout transposes and fills
MIOpen wraps around its NHWC kernels -- the timing is of the whole aten op) against csrc/conv_bwd.hip (tiled) and the data-gradient
modes of csrc/conv_fwd.hip / csrc/conv_winograd.hip.  Prints one row per distinct shape and the totals per step."""
import os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffwm_amd import ops, miopen_tuning
from ffwm_amd.flownet_eval import conv_mfma, NONE

miopen_tuning.install()
dev = "cuda"
rx = re.compile(r"\s*(\d+)x (Conv2d|ConvTranspose2d)\s+x\((\d+), (\d+), (\d+), (\d+)\)\s+-> \((\d+), (\d+), (\d+), (\d+)\)\s+k(\d+) s(\d+) p(\d+) d(\d+) dgrad=(\d) wgrad=(\d)\s+([\d.]+) GFLOP/dir\s+(.*)")


def t_us(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tot = {"w_vendor": 0.0, "w_tiled": 0.0, "d_vendor": 0.0, "d_own": 0.0}
print("%-3s %-15s %-22s %-4s k s  | wgrad: vendor   tiled (TF)  | dgrad: vendor     own (TF)   who" % ("n", "type", "input", "K"))
for line in open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "step_conv_calls.txt")):
    m = rx.match(line)
    if not m:
        continue
    n, typ = int(m.group(1)), m.group(2)
    B, C, H, W, _, K, Ho, Wo, k, s, p, d, dg, wg = [int(v) for v in m.groups()[2:16]]
    gflop, who = float(m.group(17)), m.group(18)
    if d != 1 or k not in (3, 4) or not (dg or wg):
        continue
    tr = typ == "ConvTranspose2d"
    x = torch.randn(B, C, H, W, device=dev)
    w = torch.randn(*((C, K, k, k) if tr else (K, C, k, k)), device=dev) * 0.05
    go = torch.randn(B, K, Ho, Wo, device=dev)
    row = "%3d %-15s %-22s %4d %d %d  |" % (n, typ, "(%d,%d,%d,%d)" % (B, C, H, W), K, k, s)
    if wg:
        tv = t_us(lambda: torch.ops.aten.convolution_backward(go, x, w, [K], [s, s], [p, p], [1, 1], tr, [0, 0], 1, [False, True, True]))
        rows_, gath = (x, go) if tr else (go, x)
        if ops.conv2d_wgrad_tiled_ok(rows_):
            tt = t_us(lambda: ops.conv2d_wgrad_tiled(rows_, gath, k, s, p, want_bias=not tr))
            if tr:
                tt += t_us(lambda: go.sum((0, 2, 3)))
        else:
            tt = float("nan")
        tot["w_vendor"] += n * tv
        tot["w_tiled"] += n * (tt if tt == tt else tv)
        row += " %8.1f %8.1f (%5.1f) |" % (tv, tt, gflop / tt * 1e3 if tt == tt else 0)
    else:
        row += " %8s %8s %7s |" % ("-", "-", "")
    if dg:
        tv = t_us(lambda: torch.ops.aten.convolution_backward(go, x, w, None, [s, s], [p, p], [1, 1], tr, [0, 0], 1, [True, False, False]))
        to = float("nan")
        try:
            if tr:
                to = t_us(lambda: conv_mfma(go, w, None, 2, 1, False, NONE))                       # d(input) of ConvTranspose2d(4,2,1) = Conv2d(4,2,1)
            elif s == 2 and p == 1 and H == 2 * Ho:
                to = t_us(lambda: conv_mfma(go, w, None, 2, 1, 1 if k == 4 else 2, NONE))
            elif s == 1 and k == 3 and p == 1:
                from ffwm_amd import conv as cv
                if cv._winograd_dir_ok(go, K, C):
                    to = t_us(lambda: ops.conv3x3_winograd(go, w, None, data_gradient=True))
                else:
                    to = t_us(lambda: conv_mfma(go, w, None, 1, 1, 3, NONE))
        except Exception as e:
            row += " (own dgrad failed: %s)" % str(e)[:60]
        tot["d_vendor"] += n * tv
        tot["d_own"] += n * (to if to == to else tv)
        row += " %8.1f %8.1f (%5.1f)" % (tv, to, gflop / to * 1e3 if to == to else 0)
    print(row + "  " + who[:60])
    del x, w, go
print("per step, us: weight gradients vendor %.0f / tiled %.0f;  data gradients vendor %.0f / own %.0f" % (
    tot["w_vendor"], tot["w_tiled"], tot["d_vendor"], tot["d_own"]))
