"""One rank: what does the gradient reducer cost the HOST (enqueue time per step) and the GPU (device time per step with the
host running ahead behind a held GPU)?  DDP=0 | 1 (RCCL, one rank) ; MVLT_DDP_NULL_COLLECTIVE=1 skips the collective itself."""
import os, sys, time
if os.environ.get("SET_IN_PY"): os.environ["GPU_MAX_HW_QUEUES"] = os.environ["SET_IN_PY"]      # set inside the process, before torch
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ddp = os.environ.get("DDP", "0") == "1"
if ddp:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
    sys.stdout.flush(); real = os.dup(1); os.dup2(2, 1)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import mvlt_amd as M
from mvlt_amd.ddp import GradReducer, seed_coin_flip
from mvlt_amd.train import PretrainStep, synthetic_batch
torch.manual_seed(1234)
cfg = M.MVLBertPretrainConfig(); cfg.ITM_task = True; cfg.mlm_max_labels_per_sample = None
model = M.MVLBertForPretraining(cfg).cuda().train()
M.manual_seed(4321); seed_coin_flip(5678)
use_reducer = ddp and os.environ.get("REDUCER", "1") == "1"
reducer = GradReducer(model, bucket_bytes=int(os.environ.get("MVLT_DDP_BUCKET_MB", "64")) << 20) if use_reducer else None
if ddp and not use_reducer:          # the process group and RCCL's communicator / stream exist, nothing else
    t = torch.zeros(1 << 20, device="cuda"); dist.broadcast(t, src=0); dist.all_reduce(t); torch.cuda.synchronize()
step = PretrainStep(model, reducer=reducer, world_size=1)
batch = synthetic_batch(32, 80, "cuda", 1234)
for _ in range(10): step(batch)
torch.cuda.synchronize()
N = 20
# (1) host enqueue time: the GPU is held by a long streaming job, the host queues N steps behind it
junk = torch.zeros(1 << 28, device="cuda")
a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
a.record(); junk.add_(1.0); b.record(); torch.cuda.synchronize()
n_hold = int(400.0 / a.elapsed_time(b)) + 1          # ~400 ms
a.record()
for _ in range(n_hold): junk.add_(1.0)
b.record()
t0 = time.perf_counter()
for _ in range(N): step(batch)
t_host = (time.perf_counter() - t0) / N
c.record(); torch.cuda.synchronize()
msg = (f"DDP={int(ddp)} reducer={int(bool(reducer))} GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', '-')} null={os.environ.get('MVLT_DDP_NULL_COLLECTIVE', '0')}: host enqueue {t_host * 1e3:.2f} ms per step (hold {a.elapsed_time(b):.0f} ms); "
       f"device time behind the hold {b.elapsed_time(c) / N:.3f} ms per step")
# (2) free running
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N): step(batch)
torch.cuda.synchronize()
msg += f"; free running {(time.perf_counter() - t0) / N * 1e3:.3f} ms per step"
if ddp:
    sys.stdout.flush(); os.dup2(real, 1)
print(msg, flush=True)
if ddp:
    dist.destroy_process_group()
