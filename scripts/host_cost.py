"""Host-side cost per op wrapper call (tiny shapes, no sync inside the loop)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
dt = torch.bfloat16
a = torch.randn(256, 256, device="cuda").to(dt); w = torch.randn(256, 256, device="cuda").to(dt)
bias = torch.randn(256, device="cuda"); out = torch.empty(256, 256, dtype=dt, device="cuda"); res = torch.randn(256, 256, device="cuda").to(dt)
g = torch.ones(256, device="cuda"); b = torch.zeros(256, device="cuda")
def t(f, n=3000):
    for _ in range(100): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    dt_ = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    return dt_ * 1e6
print(f"gemm(out=)                 {t(lambda: ops.gemm(a, w, out=out)):6.2f} us")
print(f"gemm()  (allocates)        {t(lambda: ops.gemm(a, w)):6.2f} us")
print(f"gemm(bias,residual,out)    {t(lambda: ops.gemm(a, w, bias=bias, residual=res, out=out)):6.2f} us")
print(f"gemm(dgrad b_kmajor)       {t(lambda: ops.gemm(a, w, b_kmajor=True, out=out)):6.2f} us")
print(f"layernorm_fwd              {t(lambda: ops.layernorm_fwd(a, g, b, 1e-5)):6.2f} us")
print(f"torch.empty                {t(lambda: torch.empty((256, 256), dtype=dt, device='cuda')):6.2f} us")
print(f"torch.add (reference op)   {t(lambda: torch.add(a, w, out=out)):6.2f} us")
