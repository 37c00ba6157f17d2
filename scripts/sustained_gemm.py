"""Does a GEMM keep its burst speed under sustained load?  (power / clock management)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
torch.manual_seed(0)
M, N, K = 3150, 3072, 768
A = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); B = (torch.randn(N, K, device="cuda") * 0.5).bfloat16()
bias = torch.randn(N, device="cuda"); out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda"); pre = torch.empty_like(out)
f = lambda: ops.gemm(A, B, bias=bias, gelu=True, save_pre=pre, out=out)
for n in (20, 200, 2000, 10000):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    print(f"{n:6d} back-to-back launches: {e0.elapsed_time(e1)/n*1e3:6.1f} us each", flush=True)
    time.sleep(0.5)
# many DIFFERENT weight matrices (cold L2 / infinity cache for the weights, like the real step)
Ws = [(torch.randn(N, K, device="cuda") * 0.5).bfloat16() for _ in range(64)]
As = [(torch.randn(M, K, device="cuda") * 0.5).bfloat16() for _ in range(64)]
outs = [torch.empty(M, N, dtype=torch.bfloat16, device="cuda") for _ in range(16)]
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(2000): ops.gemm(As[i % 64], Ws[i % 64], bias=bias, gelu=True, save_pre=pre, out=outs[i % 16])
e1.record(); torch.cuda.synchronize()
print(f"2000 launches cycling through 64 operand sets (cold caches): {e0.elapsed_time(e1)/2000*1e3:6.1f} us each", flush=True)
