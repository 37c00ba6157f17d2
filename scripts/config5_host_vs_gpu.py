"""Config #5 (Swin-B + BERT-base, B = 8, seq 128): is the step bound by the host's enqueue or by the GPU?  The GPU is held behind a
long streaming job while the host enqueues N steps; the device time of those steps (host far ahead) against the free-running
step time and the host's enqueue time per step."""
import os, sys, time, random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvlt_amd as M
from mvlt_amd.train import PretrainStep, synthetic_batch
B = int(os.environ.get("B", 8)); N = 6
cfg = M.MVLBertPretrainConfig().use_swin_base(); cfg.ITM_task = True; cfg.mlm_max_labels_per_sample = 10
model = M.MVLBertForPretraining(cfg).cuda().train()
M.manual_seed(1); random.seed(5678)
step = PretrainStep(model)
batch = synthetic_batch(B, 128, "cuda", 1234, with_lengths=True)
for _ in range(8): step(batch)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(20): step(batch)
torch.cuda.synchronize(); free = (time.time() - t0) / 20
junk = torch.zeros(1 << 28, device="cuda")
a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
a.record()
for _ in range(250): junk.add_(1.0)
b.record()
t0 = time.time()
for _ in range(N): step(batch)
enq = (time.time() - t0) / N
c.record(); torch.cuda.synchronize()
print(f"config #5 B={B}: free-running {free*1e3:.2f} ms/step; host enqueue {enq*1e3:.2f} ms/step (while the GPU was held {a.elapsed_time(b):.0f} ms); "
      f"device time with the host ahead {b.elapsed_time(c)/N:.2f} ms/step", flush=True)
