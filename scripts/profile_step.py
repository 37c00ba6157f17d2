"""One-GPU training-step timing / profiling helper (used under rocprofv3)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvlt_amd as M
from mvlt_amd.train import PretrainStep, synthetic_batch

B = int(os.environ.get("B", 32)); steps = int(os.environ.get("STEPS", 5)); warm = int(os.environ.get("WARM", 3))
cfg = M.MVLBertPretrainConfig(); cfg.ITM_task = True
cfg.mlm_max_labels_per_sample = int(os.environ.get('MLM_CAP', 10)) or None   # dataset masks <=10 tokens/sample
model = M.MVLBertForPretraining(cfg).cuda().train()
M.manual_seed(1)
step = PretrainStep(model)
batch = synthetic_batch(B, 80, "cuda", 1234, with_lengths=os.environ.get("PACK", "1") == "1")
import random; random.seed(5678)
for i in range(warm):
    l = step(batch)
torch.cuda.synchronize()
if os.environ.get("HOLD_MS"):
    # hold the GPU with a long streaming job while the host enqueues all the steps: the traced timeline then shows dependency and
    # launch gaps only, not the tracer's slower host (scripts/step_timeline.py)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    junk = torch.zeros(1 << 28, device="cuda")            # 1 GiB: one add_ moves 2 GiB (~0.4 ms)
    a.record(); junk.add_(1.0); b.record(); torch.cuda.synchronize()
    n_hold = int(float(os.environ["HOLD_MS"]) / a.elapsed_time(b)) + 1
    a.record()
    for _ in range(n_hold):
        junk.add_(1.0)
    b.record()
t0 = time.time()
for i in range(steps):
    l = step(batch)
t_enq = time.time() - t0
e_end = torch.cuda.Event(enable_timing=True); e_end.record()
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
if os.environ.get("HOLD_MS"):
    print(f"held {a.elapsed_time(b):.1f} ms; host enqueued {steps} steps in {t_enq*1e3:.1f} ms; GPU time behind the hold "
          f"{b.elapsed_time(e_end) / steps:.3f} ms per step (host ahead while the enqueue time is below the hold)", flush=True)
print(f"B={B} ms/step={dt*1e3:.2f} pairs/s={B/dt:.1f} loss={l.item():.4f} mem={torch.cuda.max_memory_allocated()/2**30:.2f}GiB", flush=True)
# host enqueue time: run steps without waiting for the GPU
torch.cuda.synchronize()
t0 = time.time()
for i in range(steps):
    l = step(batch)
t_host = (time.time() - t0) / steps
torch.cuda.synchronize()
print(f"host enqueue ms/step={t_host*1e3:.2f}", flush=True)
if os.environ.get("STAMPS"):
    # where the step's time goes on the main stream, un-traced: events around forward / backward / optimizer of the same loop as
    # PretrainStep, the GPU held first so that the host is ahead (an event record costs ~5 us of stream time: 4 per step)
    E = lambda: torch.cuda.Event(enable_timing=True)
    ns = 6
    ev = [[E() for _ in range(4)] for _ in range(ns)]
    junk = torch.zeros(1 << 28, device="cuda")
    for _ in range(200): junk.add_(1.0)
    for i in range(ns):
        ev[i][0].record()
        loss = step.model(*batch[:4], text_lengths=batch[4]) if len(batch) > 4 else step.model(*batch)
        ev[i][1].record()
        loss.backward()
        ev[i][2].record()
        step.opt.step()
        ev[i][3].record()
    torch.cuda.synchronize()
    for i in range(ns):
        f, b, o = (ev[i][j].elapsed_time(ev[i][j + 1]) for j in range(3))
        gap = ev[i][3].elapsed_time(ev[i + 1][0]) if i + 1 < ns else 0.0
        print(f"step {i}: forward {f:.3f}  backward {b:.3f}  optimizer {o:.3f}  -> next step {gap:.3f} ms", flush=True)
if os.environ.get("CPROF"):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for i in range(3): l = step(batch)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
