"""FETCH_SIZE / WRITE_SIZE calibration on known byte counts, per access width (run under tools/pmc_run.sh):
bias_act_kernel<0,4> reads and writes N bytes with 16-byte accesses, bias_act_kernel<0,1> the same bytes with 4-byte
accesses (plane size not a multiple of 4 elements selects it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffwm_amd import flownet_eval
for hw in (4096, 4095):
    h = torch.rand(64, 256, hw, 1, device="cuda")          # 268 MB (hw 4096) / 268.4 MB
    y = torch.empty_like(h)
    for _ in range(3):
        flownet_eval.bias_act(h, None, 0, y=y)
    torch.cuda.synchronize()
    print(hw, h.numel() * 4 / 1024, "KiB each way")
