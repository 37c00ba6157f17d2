"""Where a workgroup of the Swin attention backward kernel spends its time: build the library with
`make -C <pkg>/csrc EXTRA=-DSW2_TRACE` (thread 0 of every workgroup then stamps the 100 MHz wall clock at 12 points into
the buffer passed as delta_ws), run this, rebuild without the flag.  FUSE=1: with the output projection's dgrad inside the launch
(MvltAttn.dout_weight; stages 0-2)."""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops, _lib as L
torch.manual_seed(0)
B = 32
FUSE = os.environ.get("FUSE") == "1"
NAMES = ["start", "issued", "tbl+barrier", "init", "top barrier", "staged", "phase A", "barrier", "phase B", "loop end",
         "lds flush", "end"]
for st, (res, nH) in enumerate([(56, 3), (28, 6), (14, 12), (7, 24)]):
    nW = (res // 7) ** 2; nseq = B * nW; Cc = nH * 32
    qkv = (torch.randn(nseq * 49, 3 * Cc, device="cuda") * 0.5).bfloat16()
    tbl = torch.randn(169, nH, device="cuda") * 0.02
    dtbl = torch.zeros_like(tbl)
    kw = dict(bias_table=tbl, nW=nW, win_res=res, shift=3 if res > 7 else 0)
    out, lse = ops.attn_fwd(qkv, L.ATTN_SWIN, nseq, 49, nH, 32, 32 ** -0.5, **kw)
    dout = torch.randn_like(out)
    dqkv = torch.empty_like(qkv)
    buf = torch.zeros(4096 * 16, dtype=torch.int64, device="cuda")
    p = ops._attn_struct(qkv, out, lse, L.ATTN_SWIN, nseq, 49, nH, 32, 32 ** -0.5, **kw)
    p.dout, p.dqkv, p.dbias_table, p.delta_ws = ops._p(dout), ops._p(dqkv), ops._p(dtbl), ops._p(buf)
    if FUSE:
        if nH > 12: continue
        wproj = (torch.randn(Cc, Cc, device="cuda") * Cc ** -0.5).bfloat16()
        p.dout_weight = ops._p(wproj)
    for _ in range(3):
        L.check(L.lib().mvlt_attn_bwd(C.byref(p), ops._stream()), "bwd")
    torch.cuda.synchronize()
    t = buf.view(-1, 16)[:, :12].cpu().double()
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    rel = (t - t0) / 100.0            # us since the first workgroup started
    d = t[:, 1:] - t[:, :-1]
    print(f"stage {st}: {t.shape[0]} workgroups; kernel span {float(rel[:, 11].max()):.1f} us; start spread {float(rel[:, 0].max()):.1f} us")
    print("   mean us per segment: " + ", ".join(f"{NAMES[i + 1]} {float(d[:, i].mean()) / 100:.2f}" for i in range(11)))
    order = rel[:, 11].argsort()
    slow = order[-max(1, len(order) // 10):]
    print("   slowest 10% of workgroups:  " + ", ".join(f"{NAMES[i + 1]} {float(d[slow, i].mean()) / 100:.2f}" for i in range(11)))
    print("   end-time percentiles us: " + ", ".join(f"p{q}={float(rel[:, 11].quantile(q / 100)):.1f}" for q in (10, 50, 90, 99, 100)))
    ids = buf.view(-1, 16)[:, 12].cpu()[: t.shape[0]]
    xcc = (ids >> 32) & 0xf; hw = ids & 0xffffffff
    cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 0x3
    for name, key in (("xcc", xcc), ("se", se), ("cu", cu)):
        print(f"   mean end by {name}: " + " ".join(f"{int(v)}:{float(rel[key == v, 11].mean()):.1f}({int((key == v).sum())})" for v in key.unique()))
    cuid = xcc * 1000 + se * 16 + cu
    per_cu = torch.stack([rel[cuid == v, 11].max() for v in cuid.unique()])
    print(f"   distinct CUs {len(cuid.unique())}; per-CU last end: p10={float(per_cu.quantile(0.1)):.1f} p50={float(per_cu.quantile(0.5)):.1f} p90={float(per_cu.quantile(0.9)):.1f}")
    late = rel[:, 0] > 1.0
    print(f"   workgroups starting >1 us late: {int(late.sum())}; mean end of on-time WGs {float(rel[~late, 11].mean()):.1f} us")
