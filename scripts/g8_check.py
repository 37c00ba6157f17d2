"""gemm8 (8-wave ping-pong engine) vs the 4-wave kernels vs torch: correctness and time, per layout / tile / epilogue.
MVLT_G8 and MVLT_G8_TILE are read per call by the library, so both paths run in one process."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
torch.manual_seed(0)
dt = torch.bfloat16


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def setmode(g8, tile=None):
    os.environ["MVLT_G8"] = str(g8)
    if tile is None: os.environ.pop("MVLT_G8_TILE", None)
    else: os.environ["MVLT_G8_TILE"] = str(tile)


def run_single(M, N, K, bk, epi, tiles=(22, 12), mdev=None):
    A = (torch.randn(M, K, device="cuda") * 0.5).to(dt)
    B = (torch.randn((K, N) if bk else (N, K), device="cuda") * 0.5).to(dt)
    kw = {}
    bias = torch.randn(N, device="cuda")
    h = (torch.randn(M, N, device="cuda")).to(dt)
    res = (torch.randn(M, N, device="cuda")).to(dt)
    Meff = M if mdev is None else mdev
    base = A.float() @ (B.float() if bk else B.float().t())
    pre = None
    if epi == "bias": kw = dict(bias=bias); ref = base + bias
    elif epi == "gelu": kw = dict(bias=bias, gelu=True); ref = F.gelu(base + bias)
    elif epi == "gelu_pre":
        pre = torch.zeros(M, N, dtype=dt, device="cuda"); kw = dict(bias=bias, gelu=True, save_pre=pre); ref = F.gelu(base + bias)
    elif epi == "ggrad": kw = dict(mul_gelu_grad=h); hf = h.float().requires_grad_(True); F.gelu(hf).backward(base); ref = hf.grad
    elif epi == "res": kw = dict(residual=res); ref = base + res.float()
    else: ref = base
    if mdev is not None:
        kw["m_dev"] = torch.tensor([mdev], dtype=torch.int32, device="cuda")
    out = {}
    line = f"M={M:6d} N={N:5d} K={K:5d} bk={int(bk)} {epi:8s} mdev={mdev}: "
    for name, g8, tile in [("old", 0, None)] + [(f"g8/{t}", 1, t) for t in tiles] + [("g8/auto", 1, None)]:
        setmode(g8, tile)
        o = torch.zeros(M, N, dtype=dt, device="cuda")
        f = lambda: ops.gemm(A, B, b_kmajor=bk, out=o, **kw)
        f(); torch.cuda.synchronize()
        e = rel(o[:Meff], ref[:Meff])
        if mdev is not None and Meff < M:
            assert float(o[Meff:].abs().max()) == 0.0, "rows beyond m_dev were written"
        if pre is not None:
            e = max(e, rel(pre[:Meff], (base + bias)[:Meff]))
        us = timeit(f)
        line += f"{name} {us:7.1f}us err {e:.1e} | "
        assert e < 8e-3, (name, e)
    print(line, flush=True)


def run_group(widths, R, mdev=None):
    items, refs = [], []
    for i, (no, ni) in enumerate(widths):
        dy = (torch.randn(R, no, device="cuda") * 0.5).to(dt); x = (torch.randn(R, ni, device="cuda") * 0.5).to(dt)
        if mdev is not None:
            dy[mdev:] = float("nan"); x[mdev:] = float("inf")          # stale rows beyond the valid count must not matter
        dw = torch.zeros(no, ni, device="cuda"); db = torch.zeros(no, device="cuda")
        Re = R if mdev is None else mdev
        refs.append((dy[:Re].float().t() @ x[:Re].float(), dy[:Re].float().sum(0)))
        items.append((dy, x, dw, db) if mdev is None else (dy, x, dw, db, torch.tensor([mdev], dtype=torch.int32, device="cuda")))
    fl = sum(2.0 * R * no * ni for no, ni in widths)
    line = f"group {widths} R={R} mdev={mdev}: "
    for name, g8, tile in [("old", 0, None), ("g8/22", 1, 22), ("g8/12", 1, 12), ("g8/11", 1, 11), ("g8/auto", 1, None)]:
        setmode(g8, tile)
        for it in items: it[2].fill_(float("nan")); it[3].fill_(float("nan"))
        ops.wgrad_group(items); torch.cuda.synchronize()
        e = max(max(rel(it[2], r[0]), rel(it[3], r[1])) for it, r in zip(items, refs))
        us = timeit(lambda: ops.wgrad_group(items))
        line += f"{name} {us:7.1f}us {fl / us / 1e6:6.0f}TF err {e:.1e} | "
        assert e < 8e-3, (name, e)
    print(line, flush=True)


which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "fwd"):
    for epi in ["none", "bias", "gelu_pre"]:
        run_single(4192, 3072, 768, False, epi)
    run_single(3090, 3072, 768, False, "gelu_pre")
    run_single(3090, 2304, 768, False, "bias")
    run_single(4192, 3072, 768, False, "gelu_pre", mdev=3090)
    run_single(6272, 1536, 384, False, "gelu_pre")
    run_single(6272, 1152, 384, False, "bias")
    run_single(1000, 520, 192, False, "gelu")
    run_single(4096, 4096, 4096, False, "none")
if which in ("all", "dgrad"):
    run_single(4192, 3072, 768, True, "ggrad")
    run_single(3090, 3072, 768, True, "ggrad")
    run_single(4192, 3072, 768, True, "ggrad", mdev=3090)
    run_single(6272, 1536, 384, True, "ggrad")
    run_single(4192, 768, 3072, True, "res")
    run_single(1000, 520, 192, True, "none")
    run_single(4096, 4096, 4096, True, "none")
if which in ("all", "wgrad"):
    run_group([(768, 3072), (3072, 768), (768, 768), (2304, 768)], 3090)
    run_group([(768, 3072), (3072, 768), (768, 768), (2304, 768)], 4192, mdev=3090)
    run_group([(768, 3072), (3072, 768), (768, 768), (2304, 768)], 4192)
    run_group([(384, 1536), (1536, 384), (384, 384), (1152, 384)], 6272)
    run_group([(768, 3072), (3072, 768), (768, 768), (2304, 768)], 1573)
if which in ("all", "wgrad", "wgrad01"):
    run_group([(96, 384), (384, 96), (288, 96), (96, 96)], 100352)
    run_group([(192, 768), (768, 192), (576, 192), (192, 192)], 25088)
    for sp in (8, 16, 24, 32):
        os.environ["MVLT_G8_SPLIT"] = str(sp)
        print("forced split", sp)
        run_group([(96, 384), (384, 96), (288, 96), (96, 96)], 100352)
        run_group([(192, 768), (768, 192), (576, 192), (192, 192)], 25088)
    os.environ.pop("MVLT_G8_SPLIT", None)
print("g8_check OK")
