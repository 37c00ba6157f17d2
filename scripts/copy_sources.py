"""Where do the small copies / fills of a pre-training step come from?  torch.profiler with Python stacks over 2 steps;
every aten::copy_ / clone / fill_ / zero_ / cat / _to_copy is attributed to the innermost frame inside this repository."""
import os, sys, collections
import torch
from torch.profiler import profile, ProfilerActivity
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mvlt_amd as M
from mvlt_amd.train import PretrainStep, synthetic_batch
from mvlt_amd.ddp import seed_coin_flip
torch.manual_seed(0)
cfg = M.MVLBertPretrainConfig(); cfg.ITM_task = True
model = M.MVLBertForPretraining(cfg).cuda().train()
seed_coin_flip(5678)
step = PretrainStep(model)
batch = synthetic_batch(32, 80, "cuda", 1234)[:4]
for _ in range(5):
    step(batch)
torch.cuda.synchronize()
NSTEP = 2
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    for _ in range(NSTEP):
        step(batch)
    torch.cuda.synchronize()
WATCH = ("aten::copy_", "aten::clone", "aten::fill_", "aten::zero_", "aten::cat", "aten::_to_copy", "aten::contiguous",
         "aten::zeros", "aten::zeros_like", "aten::empty", "aten::empty_like", "aten::index", "aten::nonzero",
         "aten::rand", "aten::bernoulli_", "aten::add", "aten::mul", "aten::div", "aten::sum", "aten::floor_", "aten::floor")
by = collections.Counter()
for ev in prof.events():
    if ev.name not in WATCH:
        continue
    where = "?"
    for fr in ev.stack:
        if "/mvlt_amd/" in fr or "medical-vision" in fr or "/bench.py" in fr:
            where = fr.split("/")[-1]
            break
    by[(ev.name, where)] += 1
for (name, where), n in sorted(by.items(), key=lambda kv: -kv[1])[:70]:
    print(f"{n / NSTEP:7.1f} /step  {name:22s} {where}")
