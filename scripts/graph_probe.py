"""Feasibility probe: capture forward+backward of the pre-training step in a HIP graph (seeds baked)."""
import os, sys, time, random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvlt_amd as M
from mvlt_amd.train import PretrainStep, synthetic_batch

B = int(os.environ.get("B", 32)); steps = int(os.environ.get("STEPS", 10))
cfg = M.MVLBertPretrainConfig(); cfg.ITM_task = True
cfg.mlm_max_labels_per_sample = 10
model = M.MVLBertForPretraining(cfg).cuda().train()
M.manual_seed(1)
step = PretrainStep(model)
batch = synthetic_batch(B, 80, "cuda", 1234)
random.random = lambda: 0.9
for i in range(3):
    l = step(batch)
torch.cuda.synchronize()
print("eager loss", l.item(), flush=True)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for i in range(2):
        l = model(*batch); l.backward(); step.opt.step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    loss = model(*batch)
    loss.backward()
torch.cuda.synchronize()
print("captured", flush=True)
for i in range(3):
    g.replay(); step.opt.step()
torch.cuda.synchronize()
t0 = time.time()
for i in range(steps):
    g.replay(); step.opt.step()
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print(f"graph: ms/step={dt*1e3:.2f} pairs/s={B/dt:.1f} loss={loss.item():.4f}", flush=True)
