"""Swin attention backward with and without the output projection's dgrad inside the launch (MvltAttn.dout_weight), B = 32 shapes:
    python scripts/bench_swin_bwd_proj.py
prints us per (proj dgrad + attention backward) pair, per launch with the projection inside, and for the one-launch backward of the
second design (mvlt_swin_wmsa2_bwd: projection dgrad + attention backward + qkv dgrad)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvlt_amd  # noqa
from mvlt_amd import ops
from mvlt_amd._lib import ATTN_SWIN
dev = torch.device("cuda:0")
B = int(os.environ.get("B", 32))


def timed(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for res, nH in ((56, 3), (28, 6), (14, 12)):
    for shift in (0, 3):
        nW, C_ = (res // 7) ** 2, 32 * nH
        B_ = B * nW
        qkv = torch.randn(B_ * 49, 3 * C_, device=dev).bfloat16()
        table = 0.5 * torch.randn(169, nH, device=dev)
        w = (torch.randn(C_, C_, device=dev) * C_ ** -0.5).bfloat16()
        kw = dict(bias_table=table, nW=nW, win_res=res, shift=shift)
        out, lse = ops.attn_fwd(qkv, ATTN_SWIN, B_, 49, nH, 32, 32 ** -0.5, **kw)
        dy = torch.randn_like(out)
        dtab = torch.zeros_like(table)
        t_g = timed(lambda: ops.gemm(dy, w, b_kmajor=True))
        dao = ops.gemm(dy, w, b_kmajor=True)
        t_a = timed(lambda: ops.attn_bwd(dao, qkv, out, lse, ATTN_SWIN, B_, 49, nH, 32, 32 ** -0.5, dbias_table=dtab, **kw))
        t_f = timed(lambda: ops.attn_bwd(dy, qkv, out, lse, ATTN_SWIN, B_, 49, nH, 32, 32 ** -0.5, dbias_table=dtab, dout_weight=w, **kw))
        wq = (torch.randn(3 * C_, C_, device=dev) * C_ ** -0.5).bfloat16()
        dqkv = ops.attn_bwd(dy, qkv, out, lse, ATTN_SWIN, B_, 49, nH, 32, 32 ** -0.5, dbias_table=dtab, dout_weight=w, **kw)
        t_q = timed(lambda: ops.gemm(dqkv, wq, b_kmajor=True))
        t_3 = timed(lambda: ops.swin_wmsa2_bwd(dy, qkv, lse, B, res, nH, shift, w, wq, table, 32 ** -0.5, dtab)) if ops.swin_wmsa2_bwd_parts(dy.dtype, B, res, C_, nH) else float("nan")
        print(f"res {res} C {C_} shift {shift}: proj dgrad {t_g:6.1f} us + attention backward {t_a:6.1f} us = {t_g + t_a:6.1f} | proj inside {t_f:6.1f} us"
              f" | + qkv dgrad {t_q:6.1f} = {t_f + t_q:6.1f} | ONE launch (partial qkv-dgrad products) {t_3:6.1f} us", flush=True)
