"""Stand-alone launches of mvlt_swin_wmsa2_fwd (fused W-MSA, second design) at a stage's shape: for rocprofv3 passes
(scripts/pmc.py --match wmsa2) and quick timing.  STAGE=0|1|2 (default 2), B (default 32), SAVE=1 training saves, N launches."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops, _lib as L
from mvlt_amd.indexing import batched_window_maps
torch.manual_seed(0)
dt = torch.bfloat16
B = int(os.environ.get("B", 32)); st = int(os.environ.get("STAGE", 2)); N = int(os.environ.get("N", 50))
res, C, nH = [(56, 96, 3), (28, 192, 6), (14, 384, 12)][st]
rows = B * res * res
x = torch.randn(rows, C, device="cuda").to(dt)
g1, b1 = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
wqkv = (torch.randn(3 * C, C, device="cuda") * C ** -0.5).to(dt)
bqkv = torch.zeros(3 * C, device="cuda")
wproj = (torch.randn(C, C, device="cuda") * C ** -0.5).to(dt)
bproj = torch.zeros(C, device="cuda")
tbl = torch.randn(169, nH, device="cuda") * 0.02
w2n, n2w = batched_window_maps(B, res, res, 7, 3, x.device)
p = ops._wmsa_struct(x, w2n, B, res, nH, 3, g1, b1, 1e-5, wqkv, bqkv, wproj, bproj, tbl, 32 ** -0.5, None)
y, ao = torch.empty_like(x), torch.empty_like(x)
p.y, p.attn_out = y.data_ptr(), ao.data_ptr()
if os.environ.get("SAVE") == "1":
    xn = torch.empty_like(x); qkv = torch.empty((rows, 3 * C), dtype=dt, device="cuda")
    lse = torch.empty((rows // 49, nH, 49), device="cuda"); ms = torch.empty((2, rows), device="cuda")
    p.xn_win, p.qkv_win, p.lse, p.mean, p.rstd = xn.data_ptr(), qkv.data_ptr(), lse.data_ptr(), ms[0].data_ptr(), ms[1].data_ptr()
ws = ops.wmsa2_sync_ws(x.device, L.lib().mvlt_swin_wmsa2_sync_words(B, res))
lib, st_, wsp = L.lib(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.c_void_p(ws.data_ptr())
for _ in range(5):
    lib.mvlt_swin_wmsa2_fwd(ctypes.byref(p), wsp, st_)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(N):
    lib.mvlt_swin_wmsa2_fwd(ctypes.byref(p), wsp, st_)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / N * 1e3
nW = (res // 7) ** 2
flop = 2.0 * B * nW * (49 * C * 3 * C + 2 * nH * 49 * 49 * 32 + 49 * C * C)
print(f"stage {st} B={B} C={C}: wmsa2 fwd {us:.1f} us per launch ({flop / us / 1e6:.0f} TFLOP/s algorithmic), sync errors {ops.wmsa2_sync_errors()}")
