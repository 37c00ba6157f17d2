"""Phase timeline of wmsa2_fwd_kernel from in-kernel stamps (diagnostic build: make -C .../csrc EXTRA=-DW2_TRACE).
Stamps: 0 start, 1 LN tile ready, 2 qkv tiles ready (4: second sub-group), 6 attention done, 7 slice published, 8 all
groups arrived, 9 full rows in LDS, 10 projection MFMAs issued, 11 end.  Units: us (100 MHz counter)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops, _lib as L
from mvlt_amd.indexing import batched_window_maps
torch.manual_seed(0)
dt = torch.bfloat16
B = int(os.environ.get("B", 32))
for st, (res, C, nH) in enumerate([(56, 96, 3), (28, 192, 6), (14, 384, 12)]):
    if os.environ.get("STAGE") and int(os.environ["STAGE"]) != st:
        continue
    rows = B * res * res
    x = torch.randn(rows, C, device="cuda").to(dt)
    g1, b1 = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    wqkv = (torch.randn(3 * C, C, device="cuda") * C ** -0.5).to(dt)
    bqkv = torch.zeros(3 * C, device="cuda")
    wproj = (torch.randn(C, C, device="cuda") * C ** -0.5).to(dt)
    bproj = torch.zeros(C, device="cuda")
    tbl = torch.randn(169, nH, device="cuda") * 0.02
    w2n, n2w = batched_window_maps(B, res, res, 7, 3, x.device)
    save = os.environ.get("SAVE", "0") == "1"
    trace = torch.zeros(256 * 64, dtype=torch.int64, device="cuda")
    L.lib().mvlt_swin_wmsa2_trace_buffer.argtypes = [ctypes.c_void_p]
    L.lib().mvlt_swin_wmsa2_trace_buffer(trace.data_ptr())
    for _ in range(5):
        y, _ = ops.swin_wmsa2_fwd(x, w2n, B, res, nH, 3, g1, b1, 1e-5, wqkv, bqkv, wproj, bproj, tbl, 32 ** -0.5, save=save)
    torch.cuda.synchronize()
    raw = trace.view(256, 64).cpu().double()
    t = raw[:, :32] / 100.0            # us
    cyc = raw[:, 32:]
    used = t[:, 0] > 0
    t = t[used]; cyc = cyc[used]
    clk = (cyc[:, 11] - cyc[:, 0]) / (t[:, 11] - t[:, 0]) / 1e3
    print(f"   shader clock over the kernel: {float(clk.median()):.2f} GHz (median over workgroups)")
    t0 = t[:, 0].min()
    names = {16: "LN inputs arrived", 1: "LN tile", 12: "qkv MFMAs (wave 0)", 2: "qkv tiles", 13: "qkv MFMAs 2nd", 4: "qkv tiles (2nd)", 17: "score MFMAs", 18: "softmax", 19: "PV + prefetch issue", 6: "attention (barrier)", 7: "published", 8: "arrived", 9: "rows in LDS", 10: "proj MFMA", 11: "end"}
    print(f"stage {st} C={C} B={B} save={save}: {int(used.sum())} workgroups; start skew {float(t[:,0].max()-t0):.2f} us; last end {float(t[:,11].max()-t0):.2f} us")
    prev = t[:, 0]
    for k in names:
        if float(t[:, k].max()) == 0:
            continue
        d = t[:, k] - prev
        print(f"   {names[k]:>16}: +{float(d.median()):6.2f} us median (min {float(d.min()):6.2f}, max {float(d.max()):6.2f});  at {float((t[:,k]-t[:,0]).median()):6.2f} us after its start")
        prev = t[:, k]
