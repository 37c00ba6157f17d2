"""GEMM microbenchmark on the pre-training step's real shapes (B=32)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
torch.manual_seed(0)
dt = torch.bfloat16
# (name, M, N, K, a_kmajor, b_kmajor, count/step)
S = []
for st, (M, C, n) in enumerate([(100352, 96, 2), (25088, 192, 2), (6272, 384, 18), (1568, 768, 2)]):
    S += [(f"s{st} qkv fwd", M, 3 * C, C, 0, 0, n), (f"s{st} fc1 fwd", M, 4 * C, C, 0, 0, n), (f"s{st} fc2 fwd", M, C, 4 * C, 0, 0, n),
          (f"s{st} proj fwd", M, C, C, 0, 0, n),
          (f"s{st} fc2 dgrad", M, 4 * C, C, 0, 1, n), (f"s{st} fc1 dgrad", M, C, 4 * C, 0, 1, n), (f"s{st} qkv dgrad", M, C, 3 * C, 0, 1, n),
          (f"s{st} fc1 wgrad", 4 * C, C, M, 1, 1, n), (f"s{st} fc2 wgrad", C, 4 * C, M, 1, 1, n), (f"s{st} qkv wgrad", 3 * C, C, M, 1, 1, n)]
Mb = 4192
S += [("bert qkv fwd", Mb, 2304, 768, 0, 0, 12), ("bert out fwd", Mb, 768, 768, 0, 0, 12), ("bert ffn1 fwd", Mb, 3072, 768, 0, 0, 12),
      ("bert ffn2 fwd", Mb, 768, 3072, 0, 0, 12), ("bert ffn2 dgrad", Mb, 3072, 768, 0, 1, 12), ("bert ffn1 dgrad", Mb, 768, 3072, 0, 1, 12),
      ("bert qkv dgrad", Mb, 768, 2304, 0, 1, 12), ("bert ffn1 wgrad", 3072, 768, Mb, 1, 1, 12), ("bert ffn2 wgrad", 768, 3072, Mb, 1, 1, 12),
      ("bert qkv wgrad", 2304, 768, Mb, 1, 1, 12), ("bert out wgrad", 768, 768, Mb, 1, 1, 12),
      ("mlm dec fwd", 2560, 30522, 768, 0, 0, 1), ("mlm dec dgrad", 2560, 768, 30522, 0, 1, 1), ("mlm dec wgrad", 30522, 768, 2560, 1, 1, 1)]
only = sys.argv[1] if len(sys.argv) > 1 else None
tot_t = tot_f = 0.0
for name, M, N, K, ak, bk, cnt in S:
    if only and only not in name:
        continue
    A = (torch.randn((K, M) if ak else (M, K), device="cuda") * 0.5).to(dt)
    B = (torch.randn((K, N) if bk else (N, K), device="cuda") * 0.5).to(dt)
    out = torch.empty((M, N), dtype=torch.float32 if ak else dt, device="cuda")
    f = lambda: ops.gemm(A, B, a_kmajor=bool(ak), b_kmajor=bool(bk), out=out, out_f32=bool(ak))
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    fl = 2.0 * M * N * K
    tot_t += us * cnt; tot_f += fl * cnt
    print(f"{name:18s} M={M:6d} N={N:5d} K={K:6d} {us:8.1f} us  {fl/us/1e6:7.1f} TF/s  x{cnt}", flush=True)
print(f"TOTAL per step: {tot_t/1e3:.2f} ms, {tot_f/tot_t/1e6:.1f} TF/s average")
