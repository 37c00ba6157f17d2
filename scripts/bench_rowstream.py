"""Round 5: the HBM-bound Swin stage-0 / 1 products (B = 32) on the row-streaming kernel (csrc/rowstream.hip) against the tile
kernels of gemm.hip (MVLT_ROWSTREAM=0 in a second process: the switch is read once) -- microseconds and GB/s of ALGORITHMIC
bytes (activation rows in, output rows out, row operands of the epilogue, the weight once).  Usage: python bench_rowstream.py"""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
torch.manual_seed(0)
# (name, M, K, N, b_kmajor, epilogue)
SH = [("s0 fc1 fwd", 100352, 96, 384, 0, "gelu_pre"), ("s0 fc2 fwd", 100352, 384, 96, 0, "scale_res"),
      ("s0 fc2 dgrad", 100352, 96, 384, 1, "aux"), ("s0 fc1 dgrad", 100352, 384, 96, 1, ""),
      ("s0 proj dgrad", 100352, 96, 96, 1, ""), ("s0 qkv dgrad", 100352, 288, 96, 1, ""),
      ("s1 fc1 fwd", 25088, 192, 768, 0, "gelu_pre"), ("s1 fc2 dgrad", 25088, 192, 768, 1, "aux"), ("s1 proj dgrad", 25088, 192, 192, 1, ""),
      ("s1 fc2 fwd (tile kernels)", 25088, 768, 192, 0, "scale_res"), ("s1 fc1 dgrad (tile kernels)", 25088, 768, 192, 1, "")]
for name, M, K, N, bk, epi in SH:
    A = (torch.randn((M, K), device="cuda") * 0.5).to(torch.bfloat16)
    W = (torch.randn((K, N) if bk else (N, K), device="cuda") * K ** -0.5).to(torch.bfloat16)
    out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    kw, extra = {}, 0
    rnd = lambda: (torch.randn((M, N), device="cuda") * 0.5).to(torch.bfloat16)
    if epi in ("gelu_pre", "scale_res"): kw["bias"] = torch.randn(N, device="cuda")
    if epi == "gelu_pre": kw["gelu"] = True; kw["save_pre"] = torch.empty_like(out); extra = N
    if epi == "scale_res": kw["residual"] = rnd(); kw["rowscale"] = (torch.ones(M // 784 + 1, device="cuda"), 784); extra = N
    if epi == "aux": kw["mul_gelu_grad"] = rnd(); extra = N
    f = lambda: ops.gemm(A, W, b_kmajor=bool(bk), out=out, **kw)
    for _ in range(5): f()
    ts = []
    for r in range(7):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    us = statistics.median(ts)
    nbytes = 2.0 * (M * (K + N + extra) + K * N)
    print(f"{name:28s} M={M:6d} K={K:3d} N={N:3d}: {us:7.1f} us  {nbytes / us / 1e3:6.0f} GB/s algorithmic ({nbytes / 1e6:5.1f} MB)  {2.0 * M * N * K / us / 1e6:6.1f} TF/s", flush=True)
