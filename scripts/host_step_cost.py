"""Host enqueue time of one pre-training step (B=32, config #2): wall time of step() from an idle GPU WITHOUT a
synchronize at the end = the time the host needs to queue the step's ~800 launches.  Run once per binding:
    python scripts/host_step_cost.py            # native host path (csrc/host.cpp)
    MVLT_NATIVE_HOST=0 python scripts/host_step_cost.py   # Python + ctypes path"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvlt_amd as M
from mvlt_amd import ops
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
host, total = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step(batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
host.sort(); total.sort()
print(f"binding={'native (_mvlt_host.so)' if ops.NATIVE else 'python+ctypes'}: host enqueue {host[len(host)//2]:.2f} ms/step (min {host[0]:.2f}), "
      f"step from idle GPU {total[len(total)//2]:.2f} ms")
