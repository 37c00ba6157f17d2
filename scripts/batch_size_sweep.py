import sys, os, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvlt_amd as M
from mvlt_amd import ops
from mvlt_amd.train import PretrainStep, synthetic_batch
torch.manual_seed(0)
cfg = M.MVLBertPretrainConfig(); cfg.ITM_task = True; cfg.mlm_max_labels_per_sample = None
model = M.MVLBertForPretraining(cfg).cuda().train()
step = PretrainStep(model)
for B in (1, 3, 8, 16, 24, 32, 48, 64):
    batch = synthetic_batch(B, 80, "cuda", 100 + B, with_lengths=True)[:4]
    for _ in range(2): l = step(batch)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(3): l = step(batch)
    torch.cuda.synchronize(); dt = (time.time() - t) / 3
    ops.wmsa2_check(sync=True)
    v = float(l.item())
    assert v == v and 5 < v < 15, v
    print(f"B={B:3d}: {dt*1e3:7.2f} ms/step  {B/dt:8.1f} pairs/s  loss {v:.4f}", flush=True)
print("ok")
