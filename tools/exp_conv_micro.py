"""MIOpen fp32 cost of the dominant conv layers of the train step, split into forward / data gradient /
weight gradient (torch.ops.aten.convolution_backward with output masks).

    python tools/exp_conv_micro.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ffwm_amd import ops  # noqa: E402

SHAPES_195 = [(8, 195, 195, 128, 3), (8, 192, 192, 128, 3), (8, 195, 192, 128, 3), (8, 192, 195, 128, 3), (8, 3, 195, 128, 3),
              (8, 195, 3, 128, 3), (8, 256, 256, 128, 3), (8, 224, 224, 128, 3)]
SHAPES = [  # (B, Cin, Cout, H, k)
    (8, 195, 195, 128, 3), (8, 195, 195, 64, 3), (8, 128, 128, 128, 3), (8, 384, 384, 32, 3),
    (8, 195, 195, 128, 1), (8, 256, 256, 32, 3), (8, 128, 128, 64, 3), (8, 195, 256, 64, 3),
    (8, 256, 256, 16, 3), (8, 64, 64, 128, 3), (8, 512, 512, 8, 3), (8, 128, 256, 16, 3), (8, 512, 512, 16, 3),
    (8, 64, 64, 64, 3), (8, 128, 128, 32, 3),
]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def main():
    dev = torch.device("cuda", 0)
    # FFWM_MIOPEN_FIND=1: let MIOpen benchmark its solvers per shape (what the in-tree find-db encodes for the
    # layers the bench actually runs); default: MIOpen's heuristic (immediate-mode) choice
    torch.backends.cudnn.benchmark = os.environ.get("FFWM_MIOPEN_FIND", "0") == "1"
    print("%-28s %9s %9s %9s %9s   TF: fwd dgrad wgrad" % ("layer", "fwd us", "dgrad us", "wgrad us", "bias us"))
    for B, ci, co, H, k in (SHAPES_195 if os.environ.get("FFWM_SHAPES") == "195" else SHAPES):
        x = torch.randn(B, ci, H, H, device=dev)
        w = torch.randn(co, ci, k, k, device=dev) * 0.05
        b = torch.zeros(co, device=dev)
        go = torch.randn(B, co, H, H, device=dev)
        p = k // 2
        f = timeit(lambda: torch.ops.aten.convolution(x, w, b, [1, 1], [p, p], [1, 1], False, [0, 0], 1))

        def bw(mask):
            return torch.ops.aten.convolution_backward(go, x, w, [co], [1, 1], [p, p], [1, 1], False, [0, 0], 1, mask)
        d = timeit(lambda: bw([True, False, False]))
        g = timeit(lambda: bw([False, True, False]))
        bb = timeit(lambda: bw([False, False, True]))
        flop = 2.0 * B * H * H * ci * co * k * k
        mine = ""
        if k == 3 and H % 64 == 0:
            dw = torch.zeros(co, ci, 3, 3, device=dev)
            m = timeit(lambda: ops.conv3x3_wgrad(x, go, dw))
            mine = "  | mfma wgrad %8.1f us %6.1f TF (incl. launch glue)" % (m, flop / m / 1e6)
        print("%-28s %9.1f %9.1f %9.1f %9.1f   %6.1f %6.1f %6.1f%s" % (
            "%dx%d->%d @%d k%d" % (B, ci, co, H, k), f, d, g, bb, flop / f / 1e6, flop / d / 1e6, flop / g / 1e6, mine))


if __name__ == "__main__":
    main()
