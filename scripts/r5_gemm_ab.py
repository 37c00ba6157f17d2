"""Round 5: the five large BERT products (and two Swin stage-2 ones) -- this build's 4-wave kernels (default routing), the
8-wave engine with the 8-byte and the 16-byte epilogue, and the library yardstick (torch.matmul = hipBLASLt; NOT used by the
product) -- interleaved in ONE process (cdna_hip_programming.md 5.4 rule 24), median of ROUNDS rounds of 20 launches."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
torch.manual_seed(0)
ROUNDS = int(os.environ.get("ROUNDS", "5"))
SH = [("ffn-in fwd", 3072, 768, 0, ""), ("ffn-in fwd+gelu+pre", 3072, 768, 0, "g"), ("ffn-out fwd (bias)", 768, 3072, 0, "b"), ("qkv fwd", 2304, 768, 0, "b"),
      ("ffn-out dgrad", 3072, 768, 1, "a"), ("ffn-in dgrad", 768, 3072, 1, "r"), ("qkv dgrad", 768, 2304, 1, "r")]
if os.environ.get("SWIN"):
    SH = [("s2 fc1 fwd", 1536, 384, 0, "g"), ("s2 fc2 fwd", 384, 1536, 0, "r"), ("s2 fc1 dgrad", 384, 1536, 1, "r"), ("s2 fc2 dgrad", 1536, 384, 1, "a")]
MS = [int(x) for x in os.environ.get("MS", "4192,3150").split(",")]
VAR = [("4-wave", {"MVLT_G8": "0"}), ("g8 wide", {"MVLT_G8": "1", "MVLT_G8_WIDE": "1"}), ("g8 8-byte", {"MVLT_G8": "1", "MVLT_G8_WIDE": "0"})]
if os.environ.get("G8_TILE12"):
    VAR.append(("g8 128x256", {"MVLT_G8": "1", "MVLT_G8_WIDE": "1", "MVLT_G8_TILE": "12"}))


def timeit(f):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


for M in MS:
    for name, N, K, bk, epi in SH:
        A = (torch.randn((M, K), device="cuda") * 0.5).to(torch.bfloat16)
        B = (torch.randn((K, N) if bk else (N, K), device="cuda") * 0.5).to(torch.bfloat16)
        out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        kw = {}
        rnd = lambda: (torch.randn((M, N), device="cuda") * 0.5).to(torch.bfloat16)
        if "r" in epi: kw["residual"] = rnd(); kw["bias"] = torch.randn(N, device="cuda") if not bk else None
        if "a" in epi: kw["mul_gelu_grad"] = rnd()
        if "b" in epi or "g" in epi: kw["bias"] = torch.randn(N, device="cuda")
        if "g" in epi: kw["gelu"] = True; kw["save_pre"] = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        kw = {k: v for k, v in kw.items() if v is not None}
        f = lambda: ops.gemm(A, B, a_kmajor=False, b_kmajor=bool(bk), out=out, **kw)
        lib = (lambda: torch.matmul(A, B, out=out)) if bk else (lambda: torch.matmul(A, B.t(), out=out))
        ts = {v[0]: [] for v in VAR}; ts["library (plain)"] = []
        outs = {}
        for r in range(ROUNDS + 1):
            for vn, env in VAR:
                for k in ("MVLT_G8", "MVLT_G8_WIDE", "MVLT_G8_TILE"): os.environ.pop(k, None)
                os.environ.update(env)
                for _ in range(3): f()
                t = timeit(f)
                if r: ts[vn].append(t)
                if r == 0: outs[vn] = out.clone()
            for _ in range(3): lib()
            t = timeit(lib)
            if r: ts["library (plain)"].append(t)
        for k in ("MVLT_G8", "MVLT_G8_WIDE", "MVLT_G8_TILE"): os.environ.pop(k, None)
        same = all(torch.equal(outs[VAR[0][0]], o) or float((outs[VAR[0][0]].float() - o.float()).abs().max()) < 0.06 * float(o.float().abs().max()) for o in outs.values())
        line = " | ".join(f"{k} {statistics.median(v):6.1f}" for k, v in ts.items())
        print(f"M={M} {name:20s} N={N:4d} K={K:4d}: {line} us   (2MNK = {2.0*M*N*K/1e9:.1f} GF; outputs agree: {same})", flush=True)
