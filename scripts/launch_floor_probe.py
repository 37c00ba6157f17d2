"""Per-launch cost of a chain of dependent tiny kernels (one workgroup each), replayed from a HIP graph and launched
eagerly: the floor under every kernel of the step.  python scripts/launch_floor_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvlt_amd  # noqa
from mvlt_amd import ops
dev = torch.device("cuda:0")
x = torch.zeros(256, device=dev, dtype=torch.float32); y = torch.empty(256, device=dev, dtype=torch.bfloat16)
N = 1000
def chain():
    for _ in range(N // 2):
        ops.cast(x, torch.bfloat16, out=y); ops.cast(y, torch.float32, out=x)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    with ops.pin_stream(): chain()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    with ops.pin_stream(): chain()
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print(f"graph replay : {t(g.replay) * 1e3 / N:6.2f} us per dependent one-workgroup kernel")
def eager():
    with ops.pin_stream(): chain()
print(f"eager ctypes : {t(eager) * 1e3 / N:6.2f} us per launch (host-bound if > the graph figure)")
