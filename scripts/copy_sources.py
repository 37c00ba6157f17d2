"""Which host calls are behind the small device copies / fills of a pre-training step?  torch.profiler (CPU + GPU
activities) over 2 steps; every hipMemcpy* / hipMemset* runtime call is attributed to the innermost ATen / autograd op
that encloses it in time on its thread, and the GPU-side memcpy / memset records are counted by kind and size."""
import os, sys, json, collections, tempfile
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(NSTEP):
        step(batch)
    torch.cuda.synchronize()
path = os.path.join(tempfile.gettempdir(), "mvlt_copy_trace.json")
prof.export_chrome_trace(path)
ev = json.load(open(path))["traceEvents"]
ops = [e for e in ev if e.get("ph") == "X" and e.get("cat") in ("cpu_op", "user_annotation", "python_function")]
rt = [e for e in ev if e.get("ph") == "X" and e.get("cat") in ("cuda_runtime", "cuda_driver") and
      ("emcpy" in e["name"] or "emset" in e["name"])]
gpu = [e for e in ev if e.get("ph") == "X" and e.get("cat") in ("gpu_memcpy", "gpu_memset")]
by = collections.Counter()
for r in rt:
    best = None
    for o in ops:
        if o.get("tid") == r.get("tid") and o["ts"] <= r["ts"] and o["ts"] + o["dur"] >= r["ts"] + r.get("dur", 0):
            if best is None or o["dur"] < best["dur"]:
                best = o
    by[(r["name"], best["name"] if best else "(no enclosing op: native host code / Python)")] += 1
print("runtime copy / set calls per step, by enclosing op:")
for (name, op), n in sorted(by.items(), key=lambda kv: -kv[1]):
    print(f"  {n / NSTEP:6.1f}  {name:28s} <- {op}")
kinds = collections.Counter((g["name"], g.get("args", {}).get("bytes", g.get("args", {}).get("Bytes", "?"))) for g in gpu)
print("GPU-side records per step (name, bytes):")
for (name, b), n in sorted(kinds.items(), key=lambda kv: -kv[1])[:30]:
    print(f"  {n / NSTEP:6.1f}  {name}  {b}")
