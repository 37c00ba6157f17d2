"""torch.profiler view of 3 pre-training steps: which ATen ops (tiny copies / fills) sit between the mvlt kernels."""
import os, sys
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        step(batch)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="count", row_limit=45, max_name_column_width=60))
