"""Experiment: flowNetF and flowNetB (independent networks on the same input) on two HIP streams."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ffwm_amd import _lib, trainer
from ffwm_amd.trainer import part_grids

dev = torch.device("cuda", 0)
_lib.load()
t = trainer.FFWMTrainer(dev, world_size=1, seed=0)
batch = trainer.synthetic_batch(8, dev, seed=1)
t.pretrain_flow_identity(batch, steps=20)
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def forward_two_streams(self, b):
    img_S, img_F = b["img_S"], b["img_F"]
    cur = torch.cuda.current_stream(dev)
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        flow_F128, flow_F64, flow_F32 = self.flowNetF(img_S)
    with torch.cuda.stream(s2):
        self.flows_B = self.flowNetB(img_S)
    cur.wait_stream(s1); cur.wait_stream(s2)
    self.img_S_warp = self.warp(img_S, flow_F128)
    self.img_S_rec = self.warp(img_F, self.flows_B[0])
    self.fake32, self.fake64, self.fake128 = self.netG(img_S, flow=[flow_F32, flow_F64, flow_F128])
    self.img_GF128 = self.gf[128](self.fake128, img_F)
    self.parts = []
    for grid in part_grids(b["lm_F"]):
        self.parts.append((self.warp(self.img_GF128, grid), self.warp(img_F, grid)))


def run(n=12):
    for _ in range(3):
        t.step(batch, batch_increment=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        t.step(batch, batch_increment=0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print("one stream : %.2f ms" % run())
orig = trainer.FFWMTrainer.forward
trainer.FFWMTrainer.forward = forward_two_streams
print("two streams: %.2f ms" % run(), {k: round(float(v), 4) for k, v in t.loss_values().items()})
trainer.FFWMTrainer.forward = orig
print("one stream : %.2f ms" % run())
