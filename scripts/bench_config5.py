"""BASELINE config #5 on one GPU: Swin-B + BERT-base, B=8/GPU (global 64 over 8), seq128 (L=179), bf16."""
import os, sys, time, random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvlt_amd as M
from mvlt_amd.train import PretrainStep, synthetic_batch
B = int(os.environ.get("B", 8)); steps = int(os.environ.get("STEPS", 20))
cfg = M.MVLBertPretrainConfig().use_swin_base(); cfg.ITM_task = True; cfg.mlm_max_labels_per_sample = 10
model = M.MVLBertForPretraining(cfg).cuda().train()
M.manual_seed(1); random.seed(5678)
step = PretrainStep(model)
batch = synthetic_batch(B, 128, "cuda", 1234, with_lengths=True)
for _ in range(5): l = step(batch)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(steps): l = step(batch)
torch.cuda.synchronize(); dt = (time.time() - t0) / steps
n = sum(p.numel() for p in model.parameters())
print(f"config #5 (Swin-B + BERT-base, {n/1e6:.1f} M params) B={B} T=128: {dt*1e3:.2f} ms/step, {B/dt:.1f} pairs/s, loss {l.item():.3f}, "
      f"mem {torch.cuda.max_memory_allocated()/2**30:.2f} GiB", flush=True)
