"""Yardstick only (not used by the product): torch.matmul (hipBLASLt/rocBLAS) on the step's GEMM shapes."""
import torch
torch.manual_seed(0)
S = [("bert ffn1 fwd", 4192, 3072, 768, "nt"), ("bert ffn2 fwd", 4192, 768, 3072, "nt"), ("bert qkv fwd", 4192, 2304, 768, "nt"),
     ("bert out fwd", 4192, 768, 768, "nt"), ("bert ffn2 dgrad", 4192, 3072, 768, "nn"), ("bert ffn1 dgrad", 4192, 768, 3072, "nn"),
     ("bert ffn1 wgrad", 3072, 768, 4192, "tn"), ("bert qkv wgrad", 2304, 768, 4192, "tn"), ("bert out wgrad", 768, 768, 4192, "tn"),
     ("s2 fc1 fwd", 6272, 1536, 384, "nt"), ("s2 fc2 fwd", 6272, 384, 1536, "nt"), ("s2 qkv fwd", 6272, 1152, 384, "nt"),
     ("s2 proj fwd", 6272, 384, 384, "nt"), ("s2 fc1 wgrad", 1536, 384, 6272, "tn"), ("s2 fc2 wgrad", 384, 1536, 6272, "tn"),
     ("s0 fc1 fwd", 100352, 384, 96, "nt"), ("s0 fc1 wgrad", 384, 96, 100352, "tn"), ("4096^3", 4096, 4096, 4096, "nt")]
for name, M, N, K, lay in S:
    if lay == "nt":   a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16(); f = lambda: a @ b.t()
    elif lay == "nn": a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(K, N, device="cuda").bfloat16(); f = lambda: a @ b
    else:             a = torch.randn(K, M, device="cuda").bfloat16(); b = torch.randn(K, N, device="cuda").bfloat16(); f = lambda: a.t() @ b
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:16s} M={M:6d} N={N:5d} K={K:6d} {lay}: {us:7.1f} us {2.0*M*N*K/us/1e6:7.1f} TF/s", flush=True)
