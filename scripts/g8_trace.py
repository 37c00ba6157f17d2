"""In-kernel phase stamps of the 8-wave GEMM engine (diagnostic build -DG8_TRACE of csrc/gemm8.hip; scripts/g8_trace.sh builds it
into a scratch copy of the library): where do the ~15 us outside its K loop go?  Stamps of thread 0 of every workgroup (100 MHz
clock): 0 entry, 1 tile list built, 2 prologue issued, 3 first K-tile landed (first barrier), 4 K loop done, 5 epilogue issued,
6 last barrier, 7 overrun loads drained.  Prints medians over the workgroups, relative to the earliest entry stamp."""
import ctypes, os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MVLT_G8"] = "1"; os.environ["MVLT_G8_TILE"] = "22"
from mvlt_amd import ops, _lib
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = torch.zeros(256 * 16, dtype=torch.int64, device="cuda")
assert lib.mvlt_gemm8_trace_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
names = ["entry", "tile list", "prologue issued", "first K-tile landed", "K loop done", "epilogue issued", "last barrier", "drained"]
for M, N, K, epi in ((4192, 3072, 64, ""), (4192, 3072, 768, ""), (4192, 3072, 768, "g"), (3150, 3072, 768, "")):
    A = (torch.randn((M, K), device="cuda") * 0.5).to(torch.bfloat16); B = (torch.randn((N, K), device="cuda") * 0.5).to(torch.bfloat16)
    out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    kw = {}
    if epi == "g": kw = dict(bias=torch.randn(N, device="cuda"), gelu=True, save_pre=torch.empty_like(out))
    for _ in range(3): ops.gemm(A, B, out=out, **kw)
    torch.cuda.synchronize(); buf.zero_(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.gemm(A, B, out=out, **kw); e1.record(); torch.cuda.synchronize()
    t = buf.view(256, 16).cpu()
    live = t[:, 0] > 0
    t = t[live].double()
    t0 = t[:, 0].min()
    print(f"M={M} N={N} K={K} epi={epi or '-'}: {int(live.sum())} workgroups, kernel {e0.elapsed_time(e1) * 1e3:.1f} us (single launch, events)")
    for i, nm in enumerate(names):
        col = (t[:, i] - t0) / 100.0          # us
        print(f"   {i} {nm:22s} median {statistics.median(col.tolist()):6.2f} us   min {col.min():6.2f}  max {col.max():6.2f}")
