"""Graph-replayed cost of single kernels at tiny and step-sized shapes: where a kernel's fixed ~10 us goes.
python scripts/kernel_floor_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvlt_amd  # noqa
from mvlt_amd import ops
dev = torch.device("cuda:0")
dt = torch.bfloat16
N = 200
def measure(name, fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with ops.pin_stream():
            for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        with ops.pin_stream():
            for _ in range(N): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:58s} {e0.elapsed_time(e1) / 5 / N * 1e3:7.2f} us", flush=True)
def gemm_case(M, Nn, K, bk=False):
    A = (torch.randn(M, K, device=dev) * 0.1).to(dt)
    B = (torch.randn((K, Nn) if bk else (Nn, K), device=dev) * 0.1).to(dt)
    out = torch.empty(M, Nn, device=dev, dtype=dt)
    measure(f"gemm M={M} N={Nn} K={K} {'B k-major' if bk else 'row/row'}", lambda: ops.gemm(A, B, b_kmajor=bk, out=out))
for shp in [(64, 64, 64), (64, 64, 768), (128, 128, 64), (4096, 768, 64), (4096, 768, 768), (4096, 3072, 64), (4096, 3072, 768), (6272, 384, 64), (6272, 384, 384)]:
    gemm_case(*shp)
gemm_case(4096, 768, 768, True)
for rows, Cn in [(64, 768), (4096, 768), (6272, 384)]:
    x = torch.randn(rows, Cn, device=dev).to(dt); gmm = torch.ones(Cn, device=dev); b = torch.zeros(Cn, device=dev)
    measure(f"layernorm_fwd rows={rows} C={Cn}", lambda: ops.layernorm_fwd(x, gmm, b, 1e-5))
