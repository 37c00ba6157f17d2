"""Phase stamps of the one-launch Swin attention backward (mvlt_swin_wmsa2_bwd): build the library with
`make -C <pkg>/csrc EXTRA=-DWB2_TRACE`, run this, rebuild without the flag.  STAGE=0|1|2 (default 2), B=32."""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops, _lib as L
torch.manual_seed(0)
B = int(os.environ.get("B", 32))
NAMES = ["dY tile in LDS", "phase 1 done"] + [f"{n} {i}" for i in range(3) for n in ("staged", "phase A", "phase B + dbias")] + ["tile complete", "phase 3 products", "partial rows stored", "end"]
for st, (res, nH) in enumerate([(56, 3), (28, 6), (14, 12)]):
    if "STAGE" in os.environ and int(os.environ["STAGE"]) != st: continue
    nW = (res // 7) ** 2; nseq = B * nW; Cc = nH * 32
    qkv = (torch.randn(nseq * 49, 3 * Cc, device="cuda") * 0.5).bfloat16()
    tbl = torch.randn(169, nH, device="cuda") * 0.02
    dtbl = torch.zeros_like(tbl)
    kw = dict(bias_table=tbl, nW=nW, win_res=res, shift=3)
    out, lse = ops.attn_fwd(qkv, L.ATTN_SWIN, nseq, 49, nH, 32, 32 ** -0.5, **kw)
    dy = torch.randn_like(out)
    wp = (torch.randn(Cc, Cc, device="cuda") * Cc ** -0.5).bfloat16()
    wq = (torch.randn(3 * Cc, Cc, device="cuda") * Cc ** -0.5).bfloat16()
    buf = torch.zeros(512 * 32, dtype=torch.int64, device="cuda")
    lib = L.lib()
    lib.mvlt_swin_wmsa2_bwd_trace_buffer.restype = C.c_int
    lib.mvlt_swin_wmsa2_bwd_trace_buffer.argtypes = [C.c_void_p]
    assert lib.mvlt_swin_wmsa2_bwd_trace_buffer(buf.data_ptr()) == 0
    for _ in range(3):
        buf.zero_()
        ops.swin_wmsa2_bwd(dy, qkv, lse, B, res, nH, 3, wp, wq, tbl, 32 ** -0.5, dtbl)
    torch.cuda.synchronize()
    t = buf.view(-1, 32)[:, :16].cpu().double()
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    print(f"stage {st} C={Cc}: {t.shape[0]} workgroups; start skew {float((t[:, 0] - t0).max()) / 100:.2f} us; last end {float((t[:, 15] - t0).max()) / 100:.2f} us")
    prev = t[:, 0]
    for k in range(1, 16):
        d = (t[:, k] - prev) / 100
        print(f"   {NAMES[k - 1]:>18s}: + {float(d.median()):6.2f} us median (min {float(d.min()):6.2f}, max {float(d.max()):6.2f}); at {float((t[:, k] - t[:, 0]).median()) / 100:6.2f} us")
        prev = t[:, k]
    tt = buf.view(-1, 32).cpu().double()
    tt = tt[tt[:, 0] > 0]
    for a, b, name in ((3, 18, "problem 0: staged -> score / dP products issued"), (18, 19, "-> softmax / dS, images written"), (19, 4, "-> dQ products, stores, barrier"),
                       (4, 16, "phase B: barrier -> products issued"), (16, 17, "-> stores issued"), (17, 5, "-> bias-gradient diagonals")):
        d = (tt[:, b] - tt[:, a]) / 100
        print(f"      {name:>52s}: {float(d.median()):6.2f} us")
