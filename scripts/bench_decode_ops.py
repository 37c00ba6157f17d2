"""Per-op timing of the cached decode step (B=32, 2 new tokens, bf16)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
torch.manual_seed(0)
dt = torch.bfloat16
def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 64
for name, N, K in [("qkv", 2304, 768), ("out", 768, 768), ("ffn1", 3072, 768), ("ffn2", 768, 3072), ("decoder(M=32)", 30522, 768)]:
    m = 32 if "decoder" in name else M
    a = (torch.randn(m, K, device="cuda") * 0.5).to(dt); w = (torch.randn(N, K, device="cuda") * 0.5).to(dt)
    bias = torch.randn(N, device="cuda")
    out = torch.empty(m, N, dtype=dt, device="cuda")
    us = timeit(lambda: ops.gemm(a, w, bias=bias, out=out))
    print(f"gemm {name:14s} M={m} N={N} K={K}: {us:6.1f} us  weights {N*K*2/us/1e6:5.2f} TB/s", flush=True)
B, nH, hd = 32, 12, 64
for past in (51, 100, 200):
    cap = 202
    kc = torch.randn(B, nH, cap, hd, device="cuda").to(dt); vc = torch.randn(B, nH, cap, hd, device="cuda").to(dt)
    qkv = torch.randn(B * 2, 3 * nH * hd, device="cuda").to(dt)
    out = torch.empty(B * 2, nH * hd, dtype=dt, device="cuda")
    us = timeit(lambda: ops.attn_cached(qkv, kc, vc, past, 0.125, out=out))
    print(f"attn_cached past={past}: {us:6.1f} us  K/V {2*B*nH*past*hd*2/us/1e6:5.2f} TB/s", flush=True)
x = torch.randn(64, 768, device="cuda").to(dt); g = torch.ones(768, device="cuda"); b = torch.zeros(768, device="cuda")
print(f"layernorm 64x768: {timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-12, save_stats=False)):6.1f} us")
