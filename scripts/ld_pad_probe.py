"""Does the leading dimension (row stride) of GEMM operands matter?  Same products with operands that are views into
wider buffers (stride = K + pad elements): channel / bank aliasing of power-of-two-ish strides would show up here."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
torch.manual_seed(0)
dt = torch.bfloat16


def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def view(rows, cols, pad):
    buf = (torch.randn(rows, cols + pad, device="cuda") * 0.5).to(dt)
    return buf[:, :cols]


for name, M, N, K, ak, bk in [("bert ffn2 fwd", 4192, 768, 3072, 0, 0), ("bert ffn1 fwd", 4192, 3072, 768, 0, 0),
                              ("bert ffn1 dgrad", 4192, 768, 3072, 0, 1), ("bert ffn2 dgrad", 4192, 3072, 768, 0, 1),
                              ("bert ffn1 wgrad", 3072, 768, 4192, 1, 1), ("bert ffn2 wgrad", 768, 3072, 4192, 1, 1),
                              ("s2 fc2 fwd", 6272, 384, 1536, 0, 0), ("s2 fc1 wgrad", 1536, 384, 6272, 1, 1)]:
    line = f"{name:16s} M={M} N={N} K={K}: "
    for pa, pb in [(0, 0), (64, 0), (0, 64), (64, 64), (32, 32), (8, 8)]:
        A = view(K, M, pa) if ak else view(M, K, pa)
        B = view(K, N, pb) if bk else view(N, K, pb)
        out = torch.empty(M, N, dtype=torch.float32 if ak else dt, device="cuda")
        us = timeit(lambda: ops.gemm(A, B, a_kmajor=bool(ak), b_kmajor=bool(bk), out=out, out_f32=bool(ak)))
        line += f"pad({pa},{pb}) {us:6.1f} | "
    print(line, flush=True)
