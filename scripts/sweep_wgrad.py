"""Split-K sweep on the weight-gradient shapes (k-major x k-major, long reduction)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
torch.manual_seed(0)
dt = torch.bfloat16
Mb = 4192
S = [("bert qkv", 2304, 768, Mb), ("bert ffn1", 3072, 768, Mb), ("bert ffn2", 768, 3072, Mb), ("bert out", 768, 768, Mb),
     ("s2 fc1", 1536, 384, 6272), ("s2 fc2", 384, 1536, 6272), ("s2 qkv", 1152, 384, 6272), ("s2 proj", 384, 384, 6272),
     ("s0 fc1", 384, 96, 100352), ("s1 fc1", 768, 192, 25088), ("s3 fc1", 3072, 768, 1568)]
splits = [int(x) for x in os.environ.get("SPLITS", "0,1,2,3,4,6,8").split(",")]
for name, M, N, K in S:
    A = (torch.randn((K, M), device="cuda") * 0.5).to(dt)
    B = (torch.randn((K, N), device="cuda") * 0.5).to(dt)
    out = torch.empty((M, N), dtype=torch.float32, device="cuda")
    row = []
    for sp in splits:
        f = lambda: ops.gemm(A, B, a_kmajor=True, b_kmajor=True, out=out, out_f32=True, split_k=sp)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        row.append(f"sp{sp}:{us:6.1f}us/{2.0*M*N*K/us/1e6:5.0f}TF")
    print(f"{name:10s} M={M:5d} N={N:5d} K={K:6d}  " + "  ".join(row), flush=True)
