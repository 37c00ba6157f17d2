"""Fixed vs per-k-tile cost of the forward GEMM: duration over K at one-round and two-round tile counts."""
import os, sys, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops, _lib as L
dt = torch.bfloat16
def t_us(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
lib = L.lib()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for N in (3072, 768):
    for M in (3072, 4192):
        for K in (64, 256, 768, 1536, 3072):
            A = torch.randn(M, K, device="cuda").to(dt); B = torch.randn(N, K, device="cuda").to(dt)
            out = torch.empty(M, N, dtype=dt, device="cuda")
            p = L.MvltGemm()
            p.dtype, p.M, p.N, p.K = L.BF16, M, N, K
            p.A, p.lda, p.B, p.ldb, p.C, p.ldc = A.data_ptr(), K, B.data_ptr(), K, out.data_ptr(), N
            bm, bn, sp = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            lib.mvlt_gemm_plan(ctypes.byref(p), ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(sp))
            us = t_us(lambda: lib.mvlt_gemm(ctypes.byref(p), st))
            tiles = ((M + bm.value - 1) // bm.value) * ((N + bn.value - 1) // bn.value)
            print(f"N={N:5d} M={M:5d} K={K:5d} tile {bm.value}x{bn.value} tiles={tiles:4d}: {us:7.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s", flush=True)
