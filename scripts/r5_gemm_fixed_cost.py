"""Fixed cost vs per-K-tile cost of a wide-output product (M = 4192, N = 3072, plain bf16): time at K = 64 .. 3072 for the 8-wave
engine (256 x 256 tiles), the 4-wave kernels and the library yardstick (torch.matmul, not used by the product); a line fit gives
intercept (launch + first fill + epilogue + tail) and slope (us per 64-deep K-tile)."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
torch.manual_seed(0)
M, N = int(os.environ.get("M", 4192)), 3072
KS = [64, 128, 256, 512, 768, 1536, 3072]
def timeit(f):
    for _ in range(3): f()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    return statistics.median(ts)
res = {"g8": [], "4-wave": [], "library": []}
for K in KS:
    A = (torch.randn((M, K), device="cuda") * 0.5).to(torch.bfloat16); B = (torch.randn((N, K), device="cuda") * 0.5).to(torch.bfloat16)
    out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    for name, env in (("g8", {"MVLT_G8": "1", "MVLT_G8_TILE": "22"}), ("4-wave", {"MVLT_G8": "0"})):
        for k in ("MVLT_G8", "MVLT_G8_TILE"): os.environ.pop(k, None)
        os.environ.update(env)
        res[name].append(timeit(lambda: ops.gemm(A, B, out=out)))
    for k in ("MVLT_G8", "MVLT_G8_TILE"): os.environ.pop(k, None)
    res["library"].append(timeit(lambda: torch.matmul(A, B.t(), out=out)))
print(f"M={M} N={N}; K: " + " ".join(f"{k:7d}" for k in KS))
for name, ts in res.items():
    kt = [k / 64 for k in KS]
    n = len(kt); sx, sy = sum(kt), sum(ts); sxx = sum(x * x for x in kt); sxy = sum(x * y for x, y in zip(kt, ts))
    slope = (n * sxy - sx * sy) / (n * sxx - sx * sx); icpt = (sy - slope * sx) / n
    print(f"{name:8s} us: " + " ".join(f"{t:7.1f}" for t in ts) + f"   fit: {icpt:5.1f} us + {slope:5.2f} us per K-tile")
