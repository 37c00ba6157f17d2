import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
dt = torch.bfloat16
def t(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rows, C in ((6272, 384), (4192, 768), (100352, 96), (25088, 192), (1568, 768)):
    x = torch.randn(rows, C, device="cuda").to(dt); dy = torch.randn(rows, C, device="cuda").to(dt)
    g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
    dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    y, m, r, _ = ops.layernorm_fwd(x, g, b, 1e-5)
    tf = t(lambda: ops.layernorm_fwd(x, g, b, 1e-5))
    tb = t(lambda: ops.layernorm_bwd(dy, x, m, r, g, dg, db))
    mb = rows * C * 2 / 1e6
    print(f"rows={rows:6d} C={C:4d}  fwd {tf:6.1f} us ({2*mb/tf*1e-6*1e6/1e3:5.2f} TB/s)   bwd(+reduce) {tb:6.1f} us ({3*mb/tb*1e-6*1e6/1e3:5.2f} TB/s)", flush=True)
