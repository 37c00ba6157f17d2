"""Time ops.gemm on explicit shapes: SHAPES="M,N,K,ak,bk[,epi];..." (bf16, random operands).  epi: letters of r (residual),
a (x gelu'(aux): the FFN dgrads), g (bias + GELU + saved pre-activation: the FFN-in forward), b (bias)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
torch.manual_seed(0)
for spec in os.environ.get("SHAPES", "4096,4096,4096,0,0").split(";"):
    f_ = spec.split(",")
    M, N, K, ak, bk = [int(x) for x in f_[:5]]
    epi = f_[5] if len(f_) > 5 else ""
    A = (torch.randn((K, M) if ak else (M, K), device="cuda") * 0.5).to(torch.bfloat16)
    B = (torch.randn((K, N) if bk else (N, K), device="cuda") * 0.5).to(torch.bfloat16)
    out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    kw = {}
    rnd = lambda: (torch.randn((M, N), device="cuda") * 0.5).to(torch.bfloat16)
    if "r" in epi: kw["residual"] = rnd()
    if "a" in epi: kw["mul_gelu_grad"] = rnd()
    if "b" in epi or "g" in epi: kw["bias"] = torch.randn(N, device="cuda")
    if "g" in epi: kw["gelu"] = True; kw["save_pre"] = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    f = lambda: ops.gemm(A, B, a_kmajor=bool(ak), b_kmajor=bool(bk), out=out, **kw)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    ref = (A.t() if ak else A).float() @ (B if bk else B.t()).float()
    if "b" in epi or "g" in epi: ref = ref + kw["bias"]
    if "g" in epi: ref = torch.nn.functional.gelu(ref)
    if "a" in epi:
        h = kw["mul_gelu_grad"].float()
        ref = ref * (0.5 * (1 + torch.erf(h / 2 ** 0.5)) + h * torch.exp(-0.5 * h * h) / (2 * 3.141592653589793) ** 0.5)
    if "r" in epi: ref = ref + kw["residual"].float()
    err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
    print(f"M={M} N={N} K={K} ak={ak} bk={bk} epi={epi or '-'}: {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TF/s  relerr={err:.2e}", flush=True)
