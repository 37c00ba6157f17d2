import os, sys, time, torch, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvlt_amd as M
from mvlt_amd.train import PretrainStep, synthetic_batch
cfg = M.MVLBertPretrainConfig(); cfg.ITM_task = True; cfg.mlm_max_labels_per_sample = 10
model = M.MVLBertForPretraining(cfg).cuda().train()
step = PretrainStep(model); batch = synthetic_batch(32, 80, "cuda", 1)
for _ in range(3): step(batch)
torch.cuda.synchronize()
tf = tb = to = 0.0; n = 10
for _ in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); loss = model(*batch); t1 = time.perf_counter()
    loss.backward(); t2 = time.perf_counter()
    step.opt.step(); t3 = time.perf_counter()
    tf += t1 - t0; tb += t2 - t1; to += t3 - t2
print(f"host ms: forward {tf/n*1e3:.2f}  backward {tb/n*1e3:.2f}  optimizer {to/n*1e3:.2f}", flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3): loss = model(*batch)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
