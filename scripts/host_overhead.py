"""Per-call host overhead of the op wrappers (tiny problems, GPU work negligible)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops, _lib as L
dt = torch.bfloat16
A = torch.randn(64, 64, device="cuda").to(dt); B = torch.randn(64, 64, device="cuda").to(dt)
bias = torch.randn(64, device="cuda"); res = torch.randn(64, 64, device="cuda").to(dt)
g = torch.ones(64, device="cuda"); b = torch.zeros(64, device="cuda")
out32 = torch.empty(64, 64, device="cuda"); cs = torch.empty(64, device="cuda")
def bench(name, f, n=2000):
    for _ in range(50): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    dtm = (time.perf_counter() - t) / n; torch.cuda.synchronize()
    print(f"{name:34s} {dtm*1e6:7.2f} us/call", flush=True)
with ops.pin_stream():
    bench("gemm plain", lambda: ops.gemm(A, B))
    bench("gemm bias+residual", lambda: ops.gemm(A, B, bias=bias, residual=res))
    bench("gemm wgrad out= colsum", lambda: ops.gemm(A, B, a_kmajor=True, b_kmajor=True, out=out32, out_f32=True, a_colsum=cs))
    y, m, r, _ = ops.layernorm_fwd(A, g, b, 1e-5)
    bench("layernorm_fwd", lambda: ops.layernorm_fwd(A, g, b, 1e-5))
    dg, db = torch.empty(64, device="cuda"), torch.empty(64, device="cuda")
    bench("layernorm_bwd", lambda: ops.layernorm_bwd(A, A, m, r, g, dg, db))
    bench("rows_transform", lambda: ops.rows_transform(A))
    bench("torch.empty", lambda: torch.empty((64, 64), dtype=dt, device="cuda"))
    p = L.MvltGemm()
    bench("ctypes struct create", lambda: L.MvltGemm())
    lib = L.lib()
    import ctypes as C
    bench("ctypes call ws_bytes", lambda: lib.mvlt_gemm_workspace_bytes(C.byref(p)))
    bench("data_ptr x6", lambda: (A.data_ptr(), B.data_ptr(), A.data_ptr(), B.data_ptr(), A.data_ptr(), B.data_ptr()))
