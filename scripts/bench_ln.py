"""Time mvlt_layernorm_fwd/bwd at the step's shapes (B=32): python scripts/bench_ln.py
Prints us per launch and achieved GB/s of algorithmic bytes (fwd: read x, write y; bwd: read x, dy, dres, write dx)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvlt_amd  # noqa
from mvlt_amd import ops

dev = torch.device("cuda:0")
SHAPES = [(100352, 96), (25088, 192), (6272, 384), (1568, 768), (4128, 768)]
for rows, C in SHAPES:
    x = torch.randn(rows, C, device=dev).bfloat16(); dy = torch.randn_like(x); dres = torch.randn_like(x)
    g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    y, mean, rstd, _ = ops.layernorm_fwd(x, g, b, 1e-5)
    dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    q = ops.LnReduceQueue()
    def fwd(): ops.layernorm_fwd(x, g, b, 1e-5)
    def bwd(): ops.layernorm_bwd(dy, x, mean, rstd, g, dg, db, dres=dres, defer=q); q.items.clear(); q.off = 0
    for name, fn, nbytes in (("fwd", fwd, rows * C * 4), ("bwd", bwd, rows * C * 8)):
        for _ in range(5): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        print(f"rows={rows:6d} C={C:4d} {name}: {us:7.1f} us  {nbytes / us / 1e3:7.0f} GB/s", flush=True)
