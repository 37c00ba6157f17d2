"""One gemm8 configuration, a few launches (for rocprofv3 counter passes): g8_one.py rr|rk|kk M N K [tile]
(kk = the BERT-layer weight-gradient group with R = M rows)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
kind = sys.argv[1]; M, N, K = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
os.environ["MVLT_G8"] = os.environ.get("MVLT_G8", "1")
if len(sys.argv) > 5: os.environ["MVLT_G8_TILE"] = sys.argv[5]
from mvlt_amd import ops
torch.manual_seed(0)
dt = torch.bfloat16
if kind == "kk":
    ws = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
    items = [((torch.randn(M, no, device="cuda") * .5).to(dt), (torch.randn(M, ni, device="cuda") * .5).to(dt),
              torch.zeros(no, ni, device="cuda"), torch.zeros(no, device="cuda")) for no, ni in ws]
    f = lambda: ops.wgrad_group(items)
else:
    bk = kind == "rk"
    A = (torch.randn(M, K, device="cuda") * .5).to(dt)
    B = (torch.randn((K, N) if bk else (N, K), device="cuda") * .5).to(dt)
    o = torch.empty(M, N, dtype=dt, device="cuda")
    f = lambda: ops.gemm(A, B, b_kmajor=bk, out=o)
for _ in range(6): f()
torch.cuda.synchronize()
