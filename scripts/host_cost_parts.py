import os, sys, time, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops, _lib as L
dt = torch.bfloat16
a = torch.randn(256, 256, device="cuda").to(dt); w = torch.randn(256, 256, device="cuda").to(dt)
out = torch.empty(256, 256, dtype=dt, device="cuda")
lib = L.lib()
def t(f, n=5000):
    for _ in range(100): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    d = (time.perf_counter() - t0) / n; torch.cuda.synchronize(); return d * 1e6
def build():
    p = L.MvltGemm()
    p.dtype, p.M, p.N, p.K = 1, 256, 256, 256
    p.A, p.lda, p.a_kmajor = a.data_ptr(), 256, 0
    p.B, p.ldb, p.b_kmajor = w.data_ptr(), 256, 0
    p.C, p.ldc = out.data_ptr(), 256
    p.epilogue = 0; p.split_k = 0
    return p
p = build()
st = ops._stream()
print(f"struct build (ints)      {t(build):6.2f} us")
print(f"workspace_bytes          {t(lambda: lib.mvlt_gemm_workspace_bytes(C.byref(p))):6.2f} us")
print(f"mvlt_gemm call           {t(lambda: lib.mvlt_gemm(C.byref(p), st)):6.2f} us")
print(f"asserts+shape queries    {t(lambda: (a.dim() == 2 and w.dim() == 2 and a.dtype == w.dtype and a.stride(1) == 1 and w.stride(1) == 1, a.shape, w.shape, out.dtype, out.stride(-1), out.dim(), a.is_cuda, w.is_cuda)):6.2f} us")
print(f"_ld x3 + _dt + _p x3     {t(lambda: (ops._ld(a), ops._ld(w), ops._ld(out), ops._dt(a), ops._p(a), ops._p(w), ops._p(out))):6.2f} us")
print(f"full ops.gemm            {t(lambda: ops.gemm(a, w, out=out)):6.2f} us")
