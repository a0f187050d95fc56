import os, sys
sys.path.insert(0, ".")
import torch
from ffwm_amd import trainer, miopen_tuning
miopen_tuning.install()
b = trainer.synthetic_batch(8, "cuda", seed=1)
for ts in ("0", "0", "1", "1"):
    os.environ["FFWM_TWO_STREAMS"] = ts
    t = trainer.FFWMTrainer("cuda", seed=0)
    out = []
    for i in range(4):
        l = t.step(b, batch_increment=0)
        torch.cuda.synchronize()
        out.append(round(float(l["G"]), 4))
    print("two_streams", ts, out)
