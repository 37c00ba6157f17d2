"""Time ops.gemm on explicit shapes: SHAPES="M,N,K,ak,bk;..." (bf16, random operands)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
torch.manual_seed(0)
for spec in os.environ.get("SHAPES", "4096,4096,4096,0,0").split(";"):
    M, N, K, ak, bk = [int(x) for x in spec.split(",")]
    A = (torch.randn((K, M) if ak else (M, K), device="cuda") * 0.5).to(torch.bfloat16)
    B = (torch.randn((K, N) if bk else (N, K), device="cuda") * 0.5).to(torch.bfloat16)
    out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    f = lambda: ops.gemm(A, B, a_kmajor=bool(ak), b_kmajor=bool(bk), out=out)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    ref = (A.t() if ak else A).float() @ (B if bk else B.t()).float()
    err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
    print(f"M={M} N={N} K={K} ak={ak} bk={bk}: {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TF/s  relerr={err:.2e}", flush=True)
