"""Attention kernel microbenchmark at the pre-training step's shapes (B=32, bf16).
    python scripts/bench_attn.py [stage]      # stage 0..3: only that Swin stage (for counter runs), no BERT part"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops, _lib as L
torch.manual_seed(0)
dt = torch.bfloat16
B = 32


def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = 0.0
ONLY = int(sys.argv[1]) if len(sys.argv) > 1 else None
for st, (res, C, nH, nblk) in enumerate([(56, 96, 3, 2), (28, 192, 6, 2), (14, 384, 12, 18), (7, 768, 24, 2)]):
    if ONLY is not None and st != ONLY:
        continue
    nW = (res // 7) ** 2
    nseq = B * nW
    qkv = (torch.randn(nseq * 49, 3 * C, device="cuda") * 0.5).to(dt)
    tbl = torch.randn(169, nH, device="cuda") * 0.02
    dtbl = torch.zeros_like(tbl)
    for shift in ((0, 3) if res > 7 else (0,)):
        kw = dict(bias_table=tbl, nW=nW, win_res=res, shift=shift)
        out, lse = ops.attn_fwd(qkv, L.ATTN_SWIN, nseq, 49, nH, 32, 32 ** -0.5, **kw)
        dout = torch.randn_like(out)
        tf = timeit(lambda: ops.attn_fwd(qkv, L.ATTN_SWIN, nseq, 49, nH, 32, 32 ** -0.5, **kw))
        tb = timeit(lambda: ops.attn_bwd(dout, qkv, out, lse, L.ATTN_SWIN, nseq, 49, nH, 32, 32 ** -0.5, dbias_table=dtbl, **kw))
        byt_f = qkv.numel() * 2 + out.numel() * 2
        byt_b = 2 * qkv.numel() * 2 + 2 * out.numel() * 2
        cnt = nblk / (2 if res > 7 else 1)
        tot += (tf + tb) * cnt
        print(f"swin s{st} res={res} C={C} nH={nH} shift={shift}: fwd {tf:6.1f} us ({byt_f/tf/1e6:5.2f} TB/s)  bwd {tb:6.1f} us ({byt_b/tb/1e6:5.2f} TB/s)  x{cnt:g}", flush=True)
if ONLY is not None:
    sys.exit(0)
Lq, H, nH = 131, 768, 12
qkv = (torch.randn(B * Lq, 3 * H, device="cuda") * 0.5).to(dt)
ids = torch.randint(1000, 30000, (B, 80), device="cuda"); ids[:, 50:] = 0
for mode, name in ((L.ATTN_BIDIR, "bidir"), (L.ATTN_SEQ2SEQ, "seq2seq")):
    for pd in (0.0, 0.1):
        kw = dict(text_ids=ids, obj_end=50, dropout=(pd, 1234, 3))
        out, lse = ops.attn_fwd(qkv, mode, B, Lq, nH, 64, 0.125, **kw)
        dout = torch.randn_like(out)
        tf = timeit(lambda: ops.attn_fwd(qkv, mode, B, Lq, nH, 64, 0.125, **kw))
        tb = timeit(lambda: ops.attn_bwd(dout, qkv, out, lse, mode, B, Lq, nH, 64, 0.125, **kw))
        byt_f = qkv.numel() * 2 + out.numel() * 2
        byt_b = 2 * qkv.numel() * 2 + 2 * out.numel() * 2
        if pd > 0 and mode == L.ATTN_BIDIR: tot += (tf + tb) * 12
        print(f"bert {name} p={pd}: fwd {tf:6.1f} us ({byt_f/tf/1e6:5.2f} TB/s)  bwd {tb:6.1f} us ({byt_b/tb/1e6:5.2f} TB/s)  x12", flush=True)
print(f"TOTAL per step (standalone): {tot/1e3:.2f} ms")
