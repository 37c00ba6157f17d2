"""Tile probe on the packed-row BERT shapes with their real epilogues (MVLT_TILE=bm,bn overrides the plan)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
torch.manual_seed(0)
R = int(os.environ.get("R", 3150))
dt = torch.bfloat16
def t(f, n=200):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
x768 = (torch.randn(R, 768, device="cuda") * 0.5).to(dt); x3072 = (torch.randn(R, 3072, device="cuda") * 0.5).to(dt)
w_ffn1 = (torch.randn(3072, 768, device="cuda") * 0.5).to(dt); w_ffn2 = (torch.randn(768, 3072, device="cuda") * 0.5).to(dt)
w_qkv = (torch.randn(2304, 768, device="cuda") * 0.5).to(dt); w_out = (torch.randn(768, 768, device="cuda") * 0.5).to(dt)
b3072, b768, b2304 = torch.randn(3072, device="cuda"), torch.randn(768, device="cuda"), torch.randn(2304, device="cuda")
h = torch.empty(R, 3072, dtype=dt, device="cuda"); o3072 = torch.empty_like(h); o768 = torch.empty(R, 768, dtype=dt, device="cuda")
o2304 = torch.empty(R, 2304, dtype=dt, device="cuda")
res = []
res.append(("ffn1 fwd  (bias,gelu,save_pre)", t(lambda: ops.gemm(x768, w_ffn1, bias=b3072, gelu=True, save_pre=h, out=o3072))))
res.append(("ffn2 fwd  (bias,dropout,residual)", t(lambda: ops.gemm(x3072, w_ffn2, bias=b768, dropout=(0.1, 7, 3), residual=x768, out=o768))))
res.append(("qkv fwd   (bias)", t(lambda: ops.gemm(x768, w_qkv, bias=b2304, out=o2304))))
res.append(("out fwd   (bias,dropout,residual)", t(lambda: ops.gemm(x768, w_out, bias=b768, dropout=(0.1, 7, 3), residual=x768, out=o768))))
res.append(("ffn2 dgrad (x gelu')", t(lambda: ops.gemm(x768, w_ffn2, b_kmajor=True, mul_gelu_grad=h, out=o3072))))
res.append(("ffn1 dgrad (residual)", t(lambda: ops.gemm(x3072, w_ffn1, b_kmajor=True, residual=x768, out=o768))))
res.append(("out dgrad", t(lambda: ops.gemm(x768, w_out, b_kmajor=True, out=o768))))
res.append(("qkv dgrad (residual)", t(lambda: ops.gemm(o2304, w_qkv, b_kmajor=True, residual=x768, out=o768))))
print(os.environ.get("MVLT_TILE", "auto"), " ".join(f"{n.split('(')[0].strip()}={v:.1f}" for n, v in res), f"sum={sum(v for _, v in res):.1f}", flush=True)
