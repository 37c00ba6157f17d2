"""Fused W-MSA (mvlt_swin_wmsa_fwd/bwd) vs the unfused kernel sequence at the pre-training step's shapes (B=32, bf16)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops, _lib as L
from mvlt_amd.indexing import batched_window_maps
torch.manual_seed(0)
dt = torch.bfloat16
B = int(os.environ.get("B", 32))
ONLY = os.environ.get("ONLY")          # "fused" -> only the fused kernels (skips the four-launch sequences)

def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for st, (res, C, nH) in enumerate([(56, 96, 3), (28, 192, 6), (14, 384, 12)]):
    if os.environ.get("STAGE") and int(os.environ["STAGE"]) != st:
        continue
    nW = (res // 7) ** 2
    rows = B * res * res
    x = torch.randn(rows, C, device="cuda").to(dt)
    g1, b1 = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    wqkv = (torch.randn(3 * C, C, device="cuda") * C ** -0.5).to(dt)
    bqkv = torch.zeros(3 * C, device="cuda")
    wproj = (torch.randn(C, C, device="cuda") * C ** -0.5).to(dt)
    bproj = torch.zeros(C, device="cuda")
    tbl = torch.randn(169, nH, device="cuda") * 0.02
    rs = torch.full((B,), 1.1, device="cuda")
    flop = 2.0 * B * nW * (49 * C * 3 * C + 2 * nH * 49 * 49 * 32 + 49 * C * C)
    for shift in (0, 3):
        w2n, n2w = batched_window_maps(B, res, res, 7, shift, x.device)
        args = (x, w2n, B, res, nH, shift, g1, b1, 1e-5, wqkv, bqkv, wproj, bproj, tbl, 32 ** -0.5)
        # prebuilt parameter structs, bare C calls: the Python wrapper (allocations, checks) costs more than the kernel
        import ctypes
        lib, st_ = L.lib(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        pe = ops._wmsa_struct(*args, rs)
        y = torch.empty_like(x); pe.y = y.data_ptr()
        ps = ops._wmsa_struct(*args, rs)
        ps.y = y.data_ptr()
        xn, ao = torch.empty_like(x), torch.empty_like(x)
        qkv_s = torch.empty((rows, 3 * C), dtype=dt, device="cuda")
        lse_s = torch.empty((rows // 49, nH, 49), device="cuda"); ms = torch.empty((2, rows), device="cuda")
        ps.xn_win, ps.attn_out, ps.qkv_win, ps.lse = xn.data_ptr(), ao.data_ptr(), qkv_s.data_ptr(), lse_s.data_ptr()
        ps.mean, ps.rstd = ms[0].data_ptr(), ms[1].data_ptr()
        ph = ops._wmsa_struct(*args, rs)
        ph.head_split = 1
        ph.attn_out, ph.xn_win, ph.qkv_win, ph.lse = ao.data_ptr(), xn.data_ptr(), qkv_s.data_ptr(), lse_s.data_ptr()
        ph.mean, ph.rstd = ms[0].data_ptr(), ms[1].data_ptr()
        th = timeit(lambda: lib.mvlt_swin_wmsa_fwd(ctypes.byref(ph), st_), n=200)
        tp = timeit(lambda: ops.gemm(ao, wproj, bias=bproj, residual=x, rowmap=w2n, rowscale=(rs, res * res), out=y), n=50)
        print(f"   head-split (LN + qkv + attention, saves) {th:6.1f} us + proj GEMM {tp:6.1f} us (through the Python wrapper)", flush=True)
        tf_eval = timeit(lambda: lib.mvlt_swin_wmsa_fwd(ctypes.byref(pe), st_), n=200)
        tf_save = timeit(lambda: lib.mvlt_swin_wmsa_fwd(ctypes.byref(ps), st_), n=200)
        line = f"s{st} res={res} C={C} shift={shift}: fused fwd {tf_eval:6.1f} us ({flop/tf_eval/1e6:6.1f} TFLOP/s), +saves {tf_save:6.1f} us"
        if ops.swin_wmsa2_supported(dt, B, res, C, nH):
            ws = ops.wmsa2_sync_ws(x.device, lib.mvlt_swin_wmsa2_sync_words(B, res))
            wsp = ctypes.c_void_p(ws.data_ptr())
            pe.attn_out = ao.data_ptr()
            t2_eval = timeit(lambda: lib.mvlt_swin_wmsa2_fwd(ctypes.byref(pe), wsp, st_), n=200)
            t2_save = timeit(lambda: lib.mvlt_swin_wmsa2_fwd(ctypes.byref(ps), wsp, st_), n=200)
            pe.attn_out = None
            line += f" | v2 fwd {t2_eval:6.1f} us ({flop/t2_eval/1e6:6.1f} TFLOP/s), +saves {t2_save:6.1f} us (sync errors {ops.wmsa2_sync_errors()})"
        # backward: proj dgrad + attention backward + qkv dgrad, fused (one launch) vs the three launches
        if ops.swin_wmsa_bwd_supported(dt, C, nH):
            lib.mvlt_swin_wmsa_fwd(ctypes.byref(ps), st_)
            dyw = torch.randn(rows, C, device="cuda").to(dt)
            dtab = torch.zeros(169, nH, device="cuda")
            wpt, wqt = wproj.t().contiguous(), wqkv.t().contiguous()
            pb = L.MvltSwinWmsa()
            pb.dtype, pb.B, pb.res, pb.C, pb.nH, pb.shift = L.BF16, B, res, C, nH, shift
            dq_o, dx_o = torch.empty_like(qkv_s), torch.empty_like(x)
            pb.dy_win, pb.qkv_win, pb.lse, pb.wproj_t, pb.wqkv_t = dyw.data_ptr(), qkv_s.data_ptr(), lse_s.data_ptr(), wpt.data_ptr(), wqt.data_ptr()
            pb.bias_table, pb.scale, pb.dqkv, pb.dxn_win, pb.dbias_table = tbl.data_ptr(), 32 ** -0.5, dq_o.data_ptr(), dx_o.data_ptr(), dtab.data_ptr()
            tb = timeit(lambda: lib.mvlt_swin_wmsa_bwd(ctypes.byref(pb), st_), n=200)
            bflop = 2.0 * B * nW * (49 * C * C + 5 * nH * 49 * 49 * 32 + 49 * C * 3 * C)
            line += f" | fused bwd {tb:6.1f} us ({bflop/tb/1e6:6.1f} TFLOP/s)"
            if ONLY != "fused":
                def unfused_bwd():
                    dao = ops.gemm(dyw, wproj, b_kmajor=True)
                    dq = ops.attn_bwd(dao, qkv_s, ao, lse_s, L.ATTN_SWIN, B * nW, 49, nH, 32, 32 ** -0.5, dbias_table=dtab, bias_table=tbl, nW=nW, win_res=res, shift=shift)
                    return ops.gemm(dq, wqkv, b_kmajor=True)
                line += f" vs unfused bwd (3 launches) {timeit(unfused_bwd):6.1f} us"
        if ONLY != "fused":
            def unfused():
                xn, m, r, _ = ops.layernorm_fwd(x, g1, b1, 1e-5, out_rowmap=n2w)
                qkv = ops.gemm(xn, wqkv, bias=bqkv)
                ao, lse = ops.attn_fwd(qkv, L.ATTN_SWIN, B * nW, 49, nH, 32, 32 ** -0.5, bias_table=tbl, nW=nW, win_res=res, shift=shift)
                return ops.gemm(ao, wproj, bias=bproj, residual=x, rowmap=w2n, rowscale=(rs, res * res))
            tu = timeit(unfused)
            line += f" | unfused (4 launches) {tu:6.1f} us"
        print(line, flush=True)
