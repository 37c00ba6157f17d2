"""cProfile of the host side of 10 pre-training steps (where does the enqueue time go)."""
import os, sys, cProfile, pstats
import torch
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
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step(batch)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(30)

# ---- backward runs on the autograd thread (invisible to cProfile above): time its pieces by hand
import time, collections
acc = collections.defaultdict(float)
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[label] += time.perf_counter() - t0
    setattr(obj, name, g)
from mvlt_amd import swin as S, bert as Bm, model as Mo, arena as A, runtime as R, ops
wrap(S.SwinTransformer, "_backward", "swin._backward")
wrap(S.SwinTransformer, "_block_bwd", "  swin._block_bwd (incl. native call + mark)")
wrap(Bm.MVLBert, "_backward", "bert._backward")
wrap(Mo._MlmLossFn, "backward", "mlm head backward")
wrap(Mo._LinearCEFn, "backward", "itm backward")
wrap(A.Arena, "publish_grads", "arena.publish_grads")
wrap(A.Arena, "mark", "  arena.mark")
wrap(A.Arena, "begin_backward", "arena.begin_backward")
h = ops.host()
class HW:
    def __getattr__(self, n):
        f = getattr(h, n)
        def g(*a, **k):
            t0 = time.perf_counter()
            try:
                return f(*a, **k)
            finally:
                acc["  native " + n] += time.perf_counter() - t0
        return g
ops._host = HW()
tot = 0.0
for _ in range(10):
    torch.cuda.synchronize()          # idle GPU, empty queues: the host is never throttled by queue depth
    t0 = time.perf_counter()
    step(batch)
    tot += time.perf_counter() - t0
torch.cuda.synchronize()
print(f"host enqueue time {1e2*tot:.2f} ms/step (each step queued from an idle GPU)")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"{k:50s} {1e2*v:8.3f} ms/step")
