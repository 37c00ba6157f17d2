"""Kernel-level parity (GPU): every C-ABI entry point against a plain PyTorch
fp32 statement of the same op on the same seeded inputs.  Tolerances: f32 path
1e-5 (exact f32 MFMA), bf16 path: inputs are rounded to bf16 first so only the
output rounding (2^-9) and accumulation order differ."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16]


def tol(dt):
    return 2e-5 if dt == torch.float32 else 6e-3


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def rnd(shape, dt, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dt).cuda()


@pytest.fixture(scope="module")
def ops():
    from mvlt_amd import ops as o
    return o


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("ak,bk", [(False, False), (False, True), (True, True), (True, False)])
@pytest.mark.parametrize("M,N,K", [(200, 96, 48), (333, 384, 200), (130, 768, 768), (77, 30, 50), (64, 2, 768),
                                   (1000, 288, 96)])
def test_gemm_layouts(ops, dt, ak, bk, M, N, K):
    A = rnd((M, K), dt, 1)
    B = rnd((N, K), dt, 2)
    ref = A.float() @ B.float().t()
    Ain = A.t().contiguous() if ak else A
    Bin = B.t().contiguous() if bk else B
    out = ops.gemm(Ain, Bin, a_kmajor=ak, b_kmajor=bk)
    assert out.shape == (M, N)
    assert rel(out, ref) < tol(dt)


@pytest.mark.parametrize("bk", [False, True])
@pytest.mark.parametrize("M,N,K", [(3150, 3072, 128), (1000, 4096, 64), (2100, 2048, 192), (6272, 1536, 64), (4192, 768, 256),
                                   (400, 30528, 64),         # the MLM decoder on the labelled rows: 239 column tiles, eight uneven groups
                                   (2100, 2176, 128)])       # 17 column tiles
def test_gemm_column_grouped_tile_order(ops, bk, M, N, K):
    """Products with >= 128 tiles whose tile list runs through 2 / 4 / 8 column groups (gemm_dev.h:tile_coords -- which L2 sees which
    tile): every output tile written exactly once and in its place, with ragged edges and with a device-side row count."""
    A = rnd((M, K), torch.bfloat16, 3)
    B = rnd((N, K), torch.bfloat16, 4)
    ref = A.float() @ B.float().t()
    Bin = B.t().contiguous() if bk else B
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    ops.gemm(A, Bin, b_kmajor=bk, out=out)
    assert torch.isfinite(out.float()).all()
    assert rel(out, ref) < tol(torch.bfloat16)
    used = M - 333
    out2 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    ops.gemm(A, Bin, b_kmajor=bk, out=out2, m_dev=torch.tensor([used], dtype=torch.int32, device="cuda"))
    assert torch.equal(out2[:used], out[:used])
    bm = 128                                        # rows beyond the last tile that holds a valid row stay untouched
    assert torch.isnan(out2[-(-used // bm) * bm:].float()).all()


@pytest.mark.parametrize("dt", DT)
def test_gemm_asymmetric_integer_exact(ops, dt):
    """A = I-like / asymmetric B catches row<->col swaps of the MFMA C layout."""
    M, N, K = 64, 96, 64
    A = torch.zeros(M, K)
    A[torch.arange(M), torch.arange(M) % K] = 1.0
    B = (torch.arange(N * K).reshape(N, K) % 7 - 3).float() + (torch.arange(N)[:, None] % 5).float()
    out = ops.gemm(A.to(dt).cuda(), B.to(dt).cuda())
    assert torch.equal(out.float().cpu(), A @ B.t())


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("split", [0, 3, 16])
def test_gemm_splitk_wgrad_shape(ops, dt, split):
    # wgrad-like: tiny output, huge reduction
    Mtok, Nout, Kin = 5000, 96, 192
    dY = rnd((Mtok, Nout), dt, 3, 0.1)
    X = rnd((Mtok, Kin), dt, 4, 0.1)
    ref = dY.float().t() @ X.float()
    cs = torch.empty(Nout, device="cuda")
    out = ops.gemm(dY, X, a_kmajor=True, b_kmajor=True, out_f32=True, split_k=split, a_colsum=cs)
    assert out.dtype == torch.float32 and rel(out, ref) < (1e-5 if dt == torch.float32 else 2e-3)
    assert rel(cs, dY.float().sum(0)) < 1e-5        # fused bias gradient (column sums of dY)
    out2 = ops.gemm(dY, X, a_kmajor=True, b_kmajor=True, out_f32=True, split_k=split, out=out.clone(), accumulate=True)
    assert rel(out2, 2 * ref) < (1e-5 if dt == torch.float32 else 2e-3)


# ------------------------------------------------------------------ 8-wave ping-pong engine (csrc/gemm8.hip)
def _g8(monkeypatch, tile):
    """MVLT_G8=1: every eligible product goes to the engine; MVLT_G8_TILE forces the tile shape (22 = 256 x 256,
    12 = 128 x 256).  Both are read per call by the library."""
    monkeypatch.setenv("MVLT_G8", "1")
    monkeypatch.setenv("MVLT_G8_TILE", str(tile))


@pytest.mark.parametrize("tile", [22, 12])
@pytest.mark.parametrize("bk", [False, True])
@pytest.mark.parametrize("M,N,K", [(700, 520, 192), (257, 256, 128), (1000, 1032, 320)])
def test_gemm8_layouts_and_ragged_edges(ops, monkeypatch, tile, bk, M, N, K):
    """x W^T (forward) and dy W (dgrad: k-major B through transposing LDS reads) on shapes whose edges cut tiles, several
    tiles per workgroup list, vs torch in f32; plus the device-side row count (rows beyond it are not written)."""
    _g8(monkeypatch, tile)
    A = rnd((M, K), torch.bfloat16, 1, 0.5)
    B = rnd((K, N) if bk else (N, K), torch.bfloat16, 2, 0.5)
    ref = A.float() @ (B.float() if bk else B.float().t())
    out = ops.gemm(A, B, b_kmajor=bk)
    assert rel(out, ref) < tol(torch.bfloat16)
    m_dev = torch.tensor([M - 133], dtype=torch.int32, device="cuda")
    o2 = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    ops.gemm(A, B, b_kmajor=bk, out=o2, m_dev=m_dev)
    assert rel(o2[: M - 133], ref[: M - 133]) < tol(torch.bfloat16) and float(o2[M - 133:].abs().max()) == 0.0


def test_gemm8_asymmetric_integer_exact(ops, monkeypatch):
    """A = I-like, asymmetric B: catches row <-> column swaps and k-slot permutations of either operand path."""
    _g8(monkeypatch, 22)
    M, N, K = 256, 256, 128
    A = torch.zeros(M, K)
    A[torch.arange(M), torch.arange(M) % K] = 1.0
    B = (torch.arange(N * K).reshape(N, K) % 7 - 3).float() + (torch.arange(N)[:, None] % 5).float()
    out = ops.gemm(A.bfloat16().cuda(), B.bfloat16().cuda())
    assert torch.equal(out.float().cpu(), A @ B.t())
    out = ops.gemm(A.bfloat16().cuda(), B.t().contiguous().bfloat16().cuda(), b_kmajor=True)
    assert torch.equal(out.float().cpu(), A @ B.t())


@pytest.mark.parametrize("tile", [22, 12])
def test_gemm8_epilogue_classes(ops, monkeypatch, tile):
    """Every epilogue flag set the engine instantiates (forward: bias / + GELU / + saved pre-activation; dgrad: GELU'
    or residual), against torch; any other flag set must fall back to the 4-wave kernels with the same result."""
    _g8(monkeypatch, tile)
    dt = torch.bfloat16
    M, N, K = 600, 512, 192
    A, W = rnd((M, K), dt, 5, 0.5), rnd((N, K), dt, 6, 0.2)
    bias = rnd((N,), torch.float32, 7)
    res = rnd((M, N), dt, 8)
    base = A.float() @ W.float().t() + bias
    assert rel(ops.gemm(A, W, bias=bias), base) < tol(dt)
    assert rel(ops.gemm(A, W, bias=bias, gelu=True), F.gelu(base)) < tol(dt)
    pre = torch.empty((M, N), dtype=dt, device="cuda")
    out = ops.gemm(A, W, bias=bias, gelu=True, save_pre=pre)
    assert rel(pre, base) < tol(dt) and rel(out, F.gelu(base)) < tol(dt)
    # fallback class (dropout + residual is not instantiated in the engine)
    out = ops.gemm(A, W, bias=bias, dropout=(0.1, 1234, 77), residual=res)
    keep = ops.dropout_mask(M * N, 0.1, 1234, 77, A.device).view(M, N).float()
    assert rel(out, res.float() + base * keep / 0.9) < tol(dt)
    # dgrad classes
    h = rnd((M, K), dt, 9)
    dY = rnd((M, N), dt, 10, 0.5)
    Wk = rnd((N, K), dt, 12, 0.2)                  # [K_red = N, K] k-major B
    out = ops.gemm(dY, Wk, b_kmajor=True, mul_gelu_grad=h)
    hf = h.float().requires_grad_(True)
    F.gelu(hf).backward(dY.float() @ Wk.float())
    assert rel(out, hf.grad) < tol(dt)
    r2 = rnd((M, K), dt, 13)
    assert rel(ops.gemm(dY, Wk, b_kmajor=True, residual=r2), dY.float() @ Wk.float() + r2.float()) < tol(dt)


@pytest.mark.parametrize("tile", [22, 12])
def test_gemm8_weight_gradient_group_with_stale_rows(ops, monkeypatch, tile):
    """All dW_i = dY_i^T X_i of one BertLayer as ONE persistent launch of the engine (both operands k-major, f32 output,
    bias gradients by the grouped column-sum kernel), with the reduction length on the device and NaN / Inf in the rows
    beyond it (the activation buffers are allocated for the dense bound: stale rows must not matter)."""
    _g8(monkeypatch, tile)
    dt = torch.bfloat16
    R, valid = 1400, 1237
    widths = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
    k_dev = torch.tensor([valid], dtype=torch.int32, device="cuda")
    items, refs = [], []
    for i, (no, ni) in enumerate(widths):
        dy, x = rnd((R, no), dt, 10 + i, 0.5), rnd((R, ni), dt, 20 + i, 0.5)
        dy[valid:] = float("nan"); x[valid:] = float("inf")
        dw, db = torch.zeros(no, ni, device="cuda"), torch.zeros(no, device="cuda")
        items.append((dy, x, dw, db, k_dev))
        refs.append((dy[:valid].float().t() @ x[:valid].float(), dy[:valid].float().sum(0)))
    ops.wgrad_group(items)
    for (dy, x, dw, db, _), (rw, rb) in zip(items, refs):
        assert rel(dw, rw) < 2e-5 and rel(db, rb) < 2e-5


def test_gemm8_automatic_mode_takes_only_big_products(ops, monkeypatch):
    """Default (MVLT_G8 unset): mvlt_gemm_plan-independent check through results -- a 4096^3 product and a step-sized
    one both come out right whichever kernel family the heuristic picks (the choice itself is printed by
    scripts/g8_check.py; DESIGN.md section 3 says why the step's own products stay on the 4-wave kernels)."""
    monkeypatch.delenv("MVLT_G8", raising=False)
    monkeypatch.delenv("MVLT_G8_TILE", raising=False)
    for M, N, K in [(4096, 4096, 2048), (3090, 3072, 768)]:
        A, B = rnd((M, K), torch.bfloat16, 3, 0.5), rnd((N, K), torch.bfloat16, 4, 0.5)
        assert rel(ops.gemm(A, B), A.float() @ B.float().t()) < tol(torch.bfloat16)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,N,K", [(300, 192, 96),         # 96-wide tiles: 8 bytes per lane (tile_epilogue)
                                   (300, 256, 96),         # bf16: 64-wide tiles, one fragment pair per wave (tile_epilogue_wide)
                                   (4100, 1024, 96),       # bf16: 64 x 128 tiles, two pairs, a ragged last row tile
                                   (300, 252, 96)])        # N % 8 != 0: back to the 8-byte form on 128-wide tiles
def test_gemm_epilogues(ops, dt, M, N, K):
    A, W = rnd((M, K), dt, 5), rnd((N, K), dt, 6, 0.2)
    bias = rnd((N,), torch.float32, 7)
    res = rnd((M, N), dt, 8)
    base = A.float() @ W.float().t() + bias
    # bias + gelu + save_pre
    pre = torch.empty((M, N), dtype=dt, device="cuda")
    out = ops.gemm(A, W, bias=bias, gelu=True, save_pre=pre)
    assert rel(pre, base) < tol(dt) and rel(out, F.gelu(base)) < tol(dt)
    # bias + rowscale + residual with output row scatter (Swin proj)
    perm = torch.randperm(M, generator=torch.Generator().manual_seed(1)).int().cuda()
    rs = torch.tensor([0.0, 1.25, 1.25] * (M // 300 + 1), device="cuda")[: (M + 99) // 100]
    out = ops.gemm(A, W, bias=bias, rowscale=(rs, 100), residual=res, rowmap=perm)
    exp = res.float().clone()
    sc = rs[(perm.long() // 100)]
    exp[perm.long()] += base * sc[:, None]
    assert rel(out, exp) < tol(dt)
    # dropout + residual (BERT self-output): mask regenerated through mvlt_dropout_mask
    out = ops.gemm(A, W, bias=bias, dropout=(0.1, 1234, 77), residual=res)
    keep = ops.dropout_mask(M * N, 0.1, 1234, 77, A.device).view(M, N).float()
    assert 0.85 < keep.mean().item() < 0.95
    assert rel(out, res.float() + base * keep / 0.9) < tol(dt)
    # dgrad with fused GELU'
    h = rnd((M, N), dt, 9)
    dY = rnd((M, K), dt, 10)
    dA = rnd((M, K), dt, 11)
    Wt = rnd((K, N), dt, 12, 0.2)      # [K_red, N] k-major B
    out = ops.gemm(dA, Wt, b_kmajor=True, mul_gelu_grad=h)
    hf = h.float().requires_grad_(True)
    F.gelu(hf).backward(dA.float() @ Wt.float())
    assert rel(out, hf.grad) < tol(dt)


@pytest.mark.parametrize("M,N,K", [(3150, 3072, 768),        # BertLayer FFN-in of a packed B = 32 step: 20 x 24 tiles of 160 x 128
                                   (3111, 2304, 768),        # qkv; the last row tile holds 71 of its 160 rows
                                   (6272, 1536, 384)])       # Swin stage-2 Mlp.fc1: 40 x 12 tiles
def test_gemm_tile160_forward_products(ops, M, N, K):
    """Forward products that fill the chip with ONE round of 160 x 128 tiles take gemm_glds_kernel<160, 128> (72 KB of dynamic
    LDS, five 16-row fragments per wave): the plan says so, and every epilogue the model issues on these shapes equals the
    fp32 torch statement -- bias, bias + GELU + saved pre-activation, bias + dropout + residual, and a device-side row count."""
    import ctypes as C
    from mvlt_amd import _lib as L
    dt = torch.bfloat16
    A, W = rnd((M, K), dt, 41), rnd((N, K), dt, 42, K ** -0.5)
    bias = rnd((N,), torch.float32, 43)
    q = L.MvltGemm()
    q.dtype, q.M, q.N, q.K, q.lda, q.ldb, q.ldc = L.BF16, M, N, K, K, K, N
    q.A, q.B, q.C = A.data_ptr(), W.data_ptr(), A.data_ptr()
    bm, bn, sp = C.c_int(), C.c_int(), C.c_int()
    assert L.lib().mvlt_gemm_plan(C.byref(q), C.byref(bm), C.byref(bn), C.byref(sp)) == 0
    assert (bm.value, bn.value, sp.value) == (160, 128, 1)
    q.b_kmajor = 1                                   # the dgrad of the same shape stays on the smaller tiles
    assert L.lib().mvlt_gemm_plan(C.byref(q), C.byref(bm), C.byref(bn), C.byref(sp)) == 0 and bm.value != 160
    base = A.float() @ W.float().t() + bias
    out = torch.full((M, N), float("nan"), dtype=dt, device="cuda")
    ops.gemm(A, W, bias=bias, out=out)
    assert torch.isfinite(out.float()).all() and rel(out, base) < tol(dt)
    pre = torch.empty((M, N), dtype=dt, device="cuda")
    out = ops.gemm(A, W, bias=bias, gelu=True, save_pre=pre)
    assert rel(pre, base) < tol(dt) and rel(out, F.gelu(base)) < tol(dt)
    res = rnd((M, N), dt, 44)
    out = ops.gemm(A, W, bias=bias, dropout=(0.1, 99, 5), residual=res)
    keep = ops.dropout_mask(M * N, 0.1, 99, 5, A.device).view(M, N).float()
    assert rel(out, res.float() + base * keep / 0.9) < tol(dt)
    used = M - 200
    out2 = torch.full((M, N), float("nan"), dtype=dt, device="cuda")
    ops.gemm(A, W, bias=bias, out=out2, m_dev=torch.tensor([used], dtype=torch.int32, device="cuda"))
    assert rel(out2[:used], base[:used]) < tol(dt)
    assert torch.isnan(out2[-(-used // 160) * 160:].float()).all()


# (K, N, b_kmajor, epilogue): the nine products mvlt_gemm routes to the row-streaming kernel (csrc/rowstream.hip)
_ROWSTREAM = [(96, 384, False, "gelu_pre"), (96, 384, False, "gelu"), (384, 96, False, "scale_res"), (384, 96, False, "res"),
              (192, 768, False, "gelu_pre"), (96, 384, True, "aux"), (384, 96, True, ""), (96, 96, True, ""), (288, 96, True, ""),
              (192, 768, True, "aux"), (192, 192, True, "")]


@pytest.mark.parametrize("K,N,bk,epi", _ROWSTREAM)
@pytest.mark.parametrize("M", [25088, 24576 + 16 * 37 + 5])
def test_rowstream_products(ops, K, N, bk, epi, M, monkeypatch):
    """The HBM-bound Swin stage-0 / 1 products (Mlp fc1 / fc2 forward, the dgrads of fc1 / fc2 / proj / qkv:
    visual_feature_extractor.py:135-141, 231, 252) on the weight-stationary row-streaming kernel, against the fp32 torch
    statement on the same bf16 operands; M = 25088 is stage 1 of a B = 32 step, the other M is not a multiple of 16 (row
    ranges of the persistent workgroups end inside a 32-row stage, the last fragment is partial).  Every epilogue the Swin
    block issues on these shapes: bias + GELU (+ saved pre-activation), bias + DropPath row scale + residual, x gelu'(aux)."""
    monkeypatch.setenv("MVLT_ROWSTREAM", "0x1FF")          # every shape on (the default routes the stage-0 shapes only)
    dt = torch.bfloat16
    A = rnd((M, K), dt, 31)
    W = rnd((K, N) if bk else (N, K), dt, 32, K ** -0.5)
    base = A.float() @ (W.float() if bk else W.float().t())
    kw, exp, pre = {}, base, None
    if epi in ("gelu_pre", "gelu", "scale_res", "res"):
        kw["bias"] = rnd((N,), torch.float32, 33)
        exp = base + kw["bias"]
    if epi.startswith("gelu"):
        kw["gelu"] = True
        if epi == "gelu_pre":
            pre = kw["save_pre"] = torch.full((M, N), float("nan"), dtype=dt, device="cuda")
        pre_exp, exp = exp, F.gelu(exp)
    if epi in ("scale_res", "res"):
        res = kw["residual"] = rnd((M, N), dt, 34)
        if epi == "scale_res":
            rps = 784
            rs = (0.5 + (torch.arange((M + rps - 1) // rps, device="cuda") % 3).float())
            kw["rowscale"] = (rs, rps)
            exp = exp * rs[torch.arange(M, device="cuda") // rps][:, None]
        exp = exp + res.float()
    if epi == "aux":
        h = kw["mul_gelu_grad"] = rnd((M, N), dt, 35)
        hf = h.float()
        exp = base * (0.5 * (1 + torch.erf(hf / 2 ** 0.5)) + hf * torch.exp(-0.5 * hf * hf) / (2 * 3.141592653589793) ** 0.5)
    out = torch.full((M + 1, N), float("nan"), dtype=dt, device="cuda")          # a guard row behind the output
    ops.gemm(A, W, b_kmajor=bk, out=out[:M], **kw)
    torch.cuda.synchronize()
    assert bool(torch.isnan(out[M]).all())                                        # nothing written past the last row
    assert rel(out[:M], exp) < tol(dt)
    # every row, not only the norm: a workgroup's range boundary or a wrong column chunk would hide in a relative norm
    err = (out[:M].float() - exp).abs().amax(1)
    assert float(err.max()) < 0.05 * float(exp.abs().max()) + 0.05
    if pre is not None:
        assert rel(pre, pre_exp) < tol(dt)
    # the tile kernels on the same product (shape switched off): same arithmetic, another summation order over K at most
    monkeypatch.setenv("MVLT_ROWSTREAM", "0")
    out2 = torch.empty((M, N), dtype=dt, device="cuda")
    kw2 = dict(kw)
    if pre is not None:
        kw2["save_pre"] = torch.empty((M, N), dtype=dt, device="cuda")
    ops.gemm(A, W, b_kmajor=bk, out=out2, **kw2)
    assert rel(out[:M], out2) < 2e-3


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("widths,R", [([(768, 3072), (3072, 768), (768, 768), (2304, 768)], 1573),      # one BERT layer, ragged R
                                      ([(384, 1536), (1536, 384), (384, 384), (1152, 384)], 6272),      # one Swin stage-2 block
                                      ([(192, 768), (768, 192), (192, 192), (576, 192)], 1000),         # widths multiple of 96 only
                                      ([(96, 384), (384, 96)], 500),                                    # too few tiles -> one by one
                                      ([(96, 384), (384, 96), (288, 96), (96, 96)], 9001),              # Swin stage 0: few tiles, long reduction -> in-launch k-slices (atomicAdd), 96-wide swizzled tiles
                                      ([(192, 768), (768, 192), (576, 192), (192, 192)], 8200)])       # Swin stage 1, same path
def test_wgrad_group(ops, dt, widths, R):
    """mvlt_gemm_group: the weight gradients of one layer in one launch == the products one by one."""
    items, refs = [], []
    for i, (no, ni) in enumerate(widths):
        dy, x = rnd((R, no), dt, 10 + i, 0.5), rnd((R, ni), dt, 20 + i, 0.5)
        dw = torch.full((no, ni), float("nan"), device="cuda")
        db = torch.full((no,), float("nan"), device="cuda") if i != 2 else None
        items.append((dy, x, dw, db))
        refs.append((dy.float().t() @ x.float(), dy.float().sum(0)))
    ops.wgrad_group(items)
    torch.cuda.synchronize()
    for (dy, x, dw, db), (rw, rb) in zip(items, refs):
        assert rel(dw, rw) < tol(dt)
        if db is not None:
            assert rel(db, rb) < tol(dt)


@pytest.mark.parametrize("dt", DT)
def test_gemm_weight_prefetch_is_side_effect_free(ops, dt):
    """MvltGemm.prefetch / mvlt_prefetch only read: the product is bit-identical with and without, for byte ranges
    that are not multiples of a line, for every kernel family (register-staged, LDS-DMA, k-major B, ragged rows)."""
    A, W = rnd((4100, 768), dt, 1, 0.1), rnd((768, 768), dt, 2, 0.1)
    nxt = rnd((3072 * 768 + 37,), dt, 3, 0.1)
    m_dev = torch.tensor([3001], dtype=torch.int32, device="cuda")
    for kw in (dict(), dict(b_kmajor=True), dict(m_dev=m_dev)):
        ref = ops.gemm(A, W, out=torch.zeros(4100, 768, dtype=dt, device="cuda"), **kw)
        got = ops.gemm(A, W, out=torch.zeros(4100, 768, dtype=dt, device="cuda"), prefetch=nxt, **kw)
        assert torch.equal(ref, got)
    W2 = rnd((3072, 768), dt, 4, 0.1)
    assert torch.equal(ops.gemm(A, W2), ops.gemm(A, W2, prefetch=nxt[:100]))     # shorter than a line: ignored
    before = nxt.clone()
    ops.prefetch([nxt, W, W2, nxt[5:1000]])
    torch.cuda.synchronize()
    assert torch.equal(before, nxt)


@pytest.mark.parametrize("R,used", [(1573, 1000), (9001, 8300)])
def test_wgrad_group_device_row_count(ops, R, used):
    """The reduction length of a grouped launch can live on the device (ragged batches): rows at or beyond it are
    not read -- they hold NaN here -- in the one-launch path and in the k-sliced path."""
    dt = torch.bfloat16
    widths = [(768, 3072), (3072, 768), (768, 768), (2304, 768)] if R < 8192 else [(96, 384), (384, 96), (288, 96), (96, 96)]
    m_dev = torch.tensor([used], dtype=torch.int32, device="cuda")
    items, refs = [], []
    for i, (no, ni) in enumerate(widths):
        dy, x = rnd((R, no), dt, 30 + i, 0.5), rnd((R, ni), dt, 40 + i, 0.5)
        refs.append((dy[:used].float().t() @ x[:used].float(), dy[:used].float().sum(0)))
        dy[used:] = float("nan"); x[used:] = float("nan")
        items.append((dy, x, torch.full((no, ni), float("nan"), device="cuda"), torch.full((no,), float("nan"), device="cuda"), m_dev))
    ops.wgrad_group(items)
    torch.cuda.synchronize()
    for (dy, x, dw, db, _), (rw, rb) in zip(items, refs):
        assert rel(dw, rw) < tol(dt) and rel(db, rb) < tol(dt)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,N,K", [(64, 2304, 768), (64, 768, 3072), (32, 3000, 256), (5, 2, 768), (33, 30, 64), (64, 16, 32)])
def test_gemm_skinny(ops, dt, M, N, K):
    """M <= 64 products (decode step, poolers, classifier heads) take gemm_skinny_kernel: same contract."""
    import ctypes as C
    from mvlt_amd import _lib as L
    a, w = rnd((M, K), dt, 1, 0.5), rnd((N, K), dt, 2, 0.5)
    bias = rnd((N,), torch.float32, 3)
    res = rnd((M, N), dt, 4)
    p = L.MvltGemm()
    p.dtype, p.M, p.N, p.K = (L.BF16 if dt == torch.bfloat16 else L.F32), M, N, K
    bm, bn, sp = C.c_int(), C.c_int(), C.c_int()
    assert L.lib().mvlt_gemm_plan(C.byref(p), C.byref(bm), C.byref(bn), C.byref(sp)) == 0
    assert (bm.value, bn.value, sp.value) == (64, 16, 1)
    ref = a.float() @ w.float().t()
    assert rel(ops.gemm(a, w), ref) < tol(dt)
    assert rel(ops.gemm(a, w, bias=bias, residual=res), ref + bias + res.float()) < tol(dt)
    pre = torch.empty((M, N), dtype=dt, device="cuda")
    out = ops.gemm(a, w, bias=bias, gelu=True, save_pre=pre)
    assert rel(out, F.gelu(ref + bias)) < tol(dt) and rel(pre, ref + bias) < tol(dt)
    # strided A (rows of a [B, L, H] tensor: the pooler's hidden[:, 0])
    big = rnd((M, 3, K), dt, 5, 0.5)
    assert rel(ops.gemm(big[:, 1], w), big[:, 1].float() @ w.float().t()) < tol(dt)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,N,K", [(32, 30522, 768), (3, 3000, 256), (64, 17, 64)])
def test_gemm_argmax_fused(ops, dt, M, N, K):
    """mvlt_gemm_argmax: greedy pick straight from the decoder GEMM == argmax of the materialised f32 logits."""
    a, w = rnd((M, K), dt, 1, 0.5), rnd((N, K), dt, 2, 0.5)
    bias = rnd((N,), torch.float32, 3)
    idx, val = ops.gemm_argmax(a, w, bias)
    ref = a.float() @ w.float().t() + bias
    rv, ri = ref.max(-1)
    assert torch.allclose(val, rv, rtol=2e-5 if dt == torch.float32 else 1e-3, atol=1e-4)
    # near-ties may legitimately resolve differently under a different summation order: the picked logit must be the max
    assert torch.allclose(ref.gather(1, idx[:, None]).squeeze(1), rv, rtol=2e-5 if dt == torch.float32 else 1e-3, atol=1e-4)
    assert (idx == ri).float().mean() > 0.9
    w2 = w.clone(); w2[5] = w2[2]                       # exact tie between columns 2 and 5 -> first index wins
    b2 = bias.clone(); b2[5] = b2[2] = 1e4
    idx2, _ = ops.gemm_argmax(a, w2, b2)
    assert (idx2 == 2).all()


def test_gemm_bad_args(ops):
    A = torch.zeros(4, 4, device="cuda")
    with pytest.raises(AssertionError):
        ops.gemm(A, torch.zeros(4, 5, device="cuda"))
    with pytest.raises(RuntimeError):
        ops.gemm(torch.zeros(4, 4), torch.zeros(4, 4))      # CPU tensors: no fallback


@pytest.mark.parametrize("dt", DT)
def test_colsum(ops, dt):
    x = rnd((3000, 388), dt, 13)
    assert rel(ops.colsum(x), x.float().sum(0)) < 1e-4


# ------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("C", [32, 96, 192, 384, 768, 1536])
def test_layernorm_fwd_bwd(ops, dt, C):
    rows = 523
    x = rnd((rows, C), dt, 20, 2.0)
    g = (1 + 0.1 * torch.randn(C, generator=torch.Generator().manual_seed(21))).cuda()
    b = (0.1 * torch.randn(C, generator=torch.Generator().manual_seed(22))).cuda()
    y, mean, rstd, _ = ops.layernorm_fwd(x, g, b, 1e-5)
    xr = x.float().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.layer_norm(xr, (C,), gr, br, 1e-5)
    assert rel(y, yr) < tol(dt)
    dy = rnd((rows, C), dt, 23)
    dres = rnd((rows, C), dt, 24)
    yr.backward(dy.float())
    dgam, dbet = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    dx = ops.layernorm_bwd(dy, x, mean, rstd, g, dgam, dbet, dres=dres)
    assert rel(dx, xr.grad + dres.float()) < tol(dt)
    assert rel(dgam, gr.grad) < 1e-4 and rel(dbet, br.grad) < 1e-4


@pytest.mark.parametrize("C,rows", [(96, 40013), (192, 20011), (384, 6272), (768, 4192), (128, 9001), (1024, 3001)])
def test_layernorm_bwd_low_footprint_kernel(ops, C, rows):
    """The round-6 LayerNorm backward (`ln_bwd2_kernel`: bf16, buffer-descriptor row operands, 4-wave blocks) at row counts
    that make every wave walk several row groups (up to 512 blocks x 4 waves x 64 / LPR rows per pass), with the gathered
    dy (Swin window order), the residual-path gradient, the scattered + DropPath-scaled branch output and a DEVICE row
    count below the storage rows -- against the fp32 torch statement; rows beyond the device count must stay untouched."""
    dt = torch.bfloat16
    x = rnd((rows, C), dt, 220, 2.0)
    g = (1 + 0.1 * torch.randn(C, generator=torch.Generator().manual_seed(221))).cuda()
    b = (0.1 * torch.randn(C, generator=torch.Generator().manual_seed(222))).cuda()
    _, mean, rstd, _ = ops.layernorm_fwd(x, g, b, 1e-5)
    dy = rnd((rows, C), dt, 223)
    dres = rnd((rows, C), dt, 224)
    perm = torch.randperm(rows, generator=torch.Generator().manual_seed(225)).int().cuda()
    rps = (rows + 4) // 5
    rs = torch.tensor([0.5, 1.0, 0.0, 2.0, 1.25], device="cuda")
    xr = x.float().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    F.layer_norm(xr, (C,), gr, br, 1e-5).backward(dy.float()[perm.long()])           # row r receives dy[perm[r]]
    want = xr.grad + dres.float()
    dgam, dbet = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    dx, dz = ops.layernorm_bwd(dy, x, mean, rstd, g, dgam, dbet, dy_rowmap=perm, dres=dres,
                               branch=dict(rowmap=perm, rowscale=(rs, rps)))
    assert rel(dx, want) < tol(dt)
    assert rel(dgam, gr.grad) < 2e-4 and rel(dbet, br.grad) < 2e-4
    exp = torch.empty_like(want)
    exp[perm.long()] = dx.float() * rs[torch.arange(rows, device="cuda") // rps][:, None]
    assert rel(dz, exp) < tol(dt)
    # valid rows on the device: the tail keeps its sentinel, the parameter gradients see the valid rows only
    nv = rows - 777
    rows_dev = torch.tensor([nv], dtype=torch.int32, device="cuda")
    dx2 = torch.full_like(x, 7.0)
    ops.layernorm_bwd(dy, x, mean, rstd, g, dgam, dbet, dres=dres, dx=dx2, rows_dev=rows_dev)
    xr2 = x.float()[:nv].requires_grad_(True)
    gr2 = g.clone().requires_grad_(True)
    F.layer_norm(xr2, (C,), gr2, b, 1e-5).backward(dy.float()[:nv])
    assert rel(dx2[:nv], xr2.grad + dres.float()[:nv]) < tol(dt)
    assert bool((dx2[nv:] == 7.0).all())
    assert rel(dgam, gr2.grad) < 2e-4


def test_integration_md_binding_example_runs():
    """the ctypes stub printed in INTEGRATION.md is extracted, executed as is and its function is
    checked against torch (LayerNorm + window-order row scatter, visual_feature_extractor.py:356-367)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    code = md.split("<!-- cabi-example-begin")[1].split("<!-- cabi-example-end -->")[0].split("```python\n")[1].split("```")[0]
    os.environ["MVLT_REPO"] = root
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    rows, C = 2 * 56 * 56, 96
    x = rnd((rows, C), torch.bfloat16, 91)
    norm = torch.nn.LayerNorm(C).cuda()
    with torch.no_grad():
        norm.weight.add_(0.1 * torch.randn(C, device="cuda")); norm.bias.add_(0.1 * torch.randn(C, device="cuda"))
    perm = torch.randperm(rows, generator=torch.Generator().manual_seed(5)).int().cuda()
    y = ns["shifted_window_layernorm"](x, norm, perm)
    exp = torch.empty(rows, C, device="cuda")
    exp[perm.long()] = F.layer_norm(x.float(), (C,), norm.weight, norm.bias, norm.eps)
    assert rel(y, exp) < tol(torch.bfloat16)


@pytest.mark.parametrize("dt", DT)
def test_layernorm_rowmap_gelu_merge(ops, dt):
    B, H, W, Cq = 2, 8, 8, 32
    C = 4 * Cq
    x = rnd((B, H * W, Cq), dt, 30)
    g = (1 + 0.1 * torch.randn(C, generator=torch.Generator().manual_seed(31))).cuda()
    b = (0.1 * torch.randn(C, generator=torch.Generator().manual_seed(32))).cuda()
    # patch-merging gather + LN
    y, mean, rstd, _ = ops.layernorm_fwd(x, g, b, 1e-5, merge=(H, W))
    x4 = x.float().view(B, H, W, Cq).requires_grad_(True)
    cat = torch.cat([x4[:, 0::2, 0::2], x4[:, 1::2, 0::2], x4[:, 0::2, 1::2], x4[:, 1::2, 1::2]], -1).view(B, -1, C)
    yr = F.layer_norm(cat, (C,), g, b, 1e-5)
    assert y.shape == (B, 16, C) and rel(y, yr) < tol(dt)
    dy = rnd((B, 16, C), dt, 33)
    yr.backward(dy.float())
    dgam, dbet = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    dx = ops.layernorm_bwd(dy, x, mean, rstd, g, dgam, dbet, merge=(H, W))
    assert rel(dx, x4.grad.view(B, H * W, Cq)) < tol(dt)
    # row scatter + fused GELU
    rows = 200
    x2 = rnd((rows, C), dt, 34)
    perm = torch.randperm(rows, generator=torch.Generator().manual_seed(2)).int().cuda()
    y2, mean2, rstd2, ypre = ops.layernorm_fwd(x2, g, b, 1e-5, out_rowmap=perm, gelu=True, save_pre=True)
    x2r = x2.float().requires_grad_(True)
    ln = F.layer_norm(x2r, (C,), g, b, 1e-5)
    exp = torch.empty_like(ln)
    exp[perm.long()] = F.gelu(ln)
    assert rel(y2, exp) < tol(dt)
    dy2 = rnd((rows, C), dt, 35)
    exp.backward(dy2.float())
    dx2 = ops.layernorm_bwd(dy2, x2, mean2, rstd2, g, dgam, dbet, dy_rowmap=perm, y_pre=ypre)
    assert rel(dx2, x2r.grad) < tol(dt) * 2


# ------------------------------------------------------------------ attention
def swin_ref(qkv, table, nW, res, shift, nH, scale):
    from oracle import mvlt_oracle as O
    B_ = qkv.shape[0] // 49
    hd = qkv.shape[1] // (3 * nH)
    q, k, v = qkv.view(B_, 49, 3, nH, hd).permute(2, 0, 3, 1, 4)
    att = (q * scale) @ k.transpose(-1, -2)
    idx = O.relative_position_index(7).to(qkv.device)
    att = att + table[idx.view(-1)].view(49, 49, nH).permute(2, 0, 1)[None]
    if shift:
        m = O.shift_attn_mask(res, res, 7, shift).to(qkv.device)
        att = (att.view(B_ // nW, nW, nH, 49, 49) + m[None, :, None]).view(B_, nH, 49, 49)
    return (att.softmax(-1) @ v).transpose(1, 2).reshape(B_ * 49, nH * hd)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("res,nH,shift", [(14, 12, 3), (14, 12, 0), (28, 6, 3), (56, 3, 3), (7, 24, 0)])
def test_swin_attention(ops, dt, res, nH, shift):
    from mvlt_amd._lib import ATTN_SWIN
    nW = (res // 7) ** 2
    B_ = 2 * nW
    qkv = rnd((B_ * 49, 3 * nH * 32), dt, 40)
    table = (0.5 * torch.randn(169, nH, generator=torch.Generator().manual_seed(41))).cuda()
    scale = 32 ** -0.5
    out, lse = ops.attn_fwd(qkv, ATTN_SWIN, B_, 49, nH, 32, scale, bias_table=table, nW=nW, win_res=res, shift=shift)
    qr = qkv.float().requires_grad_(True)
    tr = table.clone().requires_grad_(True)
    ref = swin_ref(qr, tr, nW, res, shift, nH, scale)
    assert rel(out, ref) < tol(dt)
    dout = rnd(out.shape, dt, 42)
    ref.backward(dout.float())
    dtab = torch.zeros_like(table)
    dqkv = ops.attn_bwd(dout, qkv, out, lse, ATTN_SWIN, B_, 49, nH, 32, scale, dbias_table=dtab, bias_table=table,
                        nW=nW, win_res=res, shift=shift)
    assert rel(dqkv, qr.grad) < tol(dt) * 3
    assert rel(dtab, tr.grad) < (1e-4 if dt == torch.float32 else 2e-2)


@pytest.mark.parametrize("res,nH,shift,B", [(14, 12, 3, 24), (14, 12, 0, 32), (28, 6, 3, 12), (56, 3, 3, 6), (56, 3, 0, 32),
                                            (7, 24, 0, 32), (14, 16, 3, 8)])
def test_swin_attention_backward_full_batches(ops, res, nH, shift, B):
    """bf16 Swin attention backward (the scores-once kernel) at the step's batch sizes: several windows per workgroup,
    every window kind of a shifted block, fewer windows than workgroup slots; against the fp32 torch statement of
    visual_feature_extractor.py:224-251, globally and per window."""
    from mvlt_amd._lib import ATTN_SWIN
    dt = torch.bfloat16
    nW = (res // 7) ** 2
    B_ = B * nW
    qkv = rnd((B_ * 49, 3 * nH * 32), dt, 43)
    table = (0.5 * torch.randn(169, nH, generator=torch.Generator().manual_seed(44))).cuda()
    scale = 32 ** -0.5
    kw = dict(bias_table=table, nW=nW, win_res=res, shift=shift)
    out, lse = ops.attn_fwd(qkv, ATTN_SWIN, B_, 49, nH, 32, scale, **kw)
    qr = qkv.float().requires_grad_(True)
    tr = table.clone().requires_grad_(True)
    ref = swin_ref(qr, tr, nW, res, shift, nH, scale)
    dout = rnd(out.shape, dt, 45)
    ref.backward(dout.float())
    dtab = torch.zeros_like(table)
    dqkv = ops.attn_bwd(dout, qkv, out, lse, ATTN_SWIN, B_, 49, nH, 32, scale, dbias_table=dtab, **kw)
    assert rel(dqkv, qr.grad) < tol(dt) * 3
    # per window: a wave that mixed up two windows would still have a small global error at B = 32
    e = (dqkv.float() - qr.grad).view(B_, -1).norm(dim=1) / (qr.grad.view(B_, -1).norm(dim=1) + 1e-30)
    assert float(e.max()) < tol(dt) * 4, int(e.argmax())
    assert rel(dtab, tr.grad) < 2e-2


@pytest.mark.parametrize("res,nH,shift,B", [(14, 12, 3, 24), (14, 12, 0, 32), (28, 6, 3, 12), (28, 6, 0, 3), (56, 3, 3, 6), (56, 3, 0, 32)])
def test_swin_attention_backward_with_projection_dgrad(ops, res, nH, shift, B):
    """MvltAttn.dout_weight: the output projection's dgrad inside the Swin attention backward (WindowAttention.proj,
    visual_feature_extractor.py:252 backward) == the separate product followed by the plain launch, and both == the fp32 torch
    statement with the projection included, per window as well; unsupported shapes are refused, not mis-computed."""
    from mvlt_amd._lib import ATTN_SWIN
    dt = torch.bfloat16
    nW = (res // 7) ** 2
    B_, C_ = B * nW, 32 * nH
    qkv = rnd((B_ * 49, 3 * C_), dt, 53)
    table = (0.5 * torch.randn(169, nH, generator=torch.Generator().manual_seed(54))).cuda()
    wproj = rnd((C_, C_), dt, 56, C_ ** -0.5)
    scale = 32 ** -0.5
    kw = dict(bias_table=table, nW=nW, win_res=res, shift=shift)
    out, lse = ops.attn_fwd(qkv, ATTN_SWIN, B_, 49, nH, 32, scale, **kw)
    dy = rnd(out.shape, dt, 55)
    qr = qkv.float().requires_grad_(True)
    tr = table.clone().requires_grad_(True)
    (swin_ref(qr, tr, nW, res, shift, nH, scale) @ wproj.float().t()).backward(dy.float())
    dtab_a, dtab_b = torch.zeros_like(table), torch.zeros_like(table)
    dao = ops.gemm(dy, wproj, b_kmajor=True)
    two = ops.attn_bwd(dao, qkv, out, lse, ATTN_SWIN, B_, 49, nH, 32, scale, dbias_table=dtab_a, **kw)
    one = ops.attn_bwd(dy, qkv, out, lse, ATTN_SWIN, B_, 49, nH, 32, scale, dbias_table=dtab_b, dout_weight=wproj, **kw)
    torch.cuda.synchronize()
    assert torch.isfinite(one.float()).all()
    assert rel(one, two) < 2e-3 and rel(dtab_b, dtab_a) < 2e-3          # same bf16 dO up to the summation order of its f32 sums
    assert rel(one, qr.grad) < tol(dt) * 3
    e = (one.float() - qr.grad).view(B_, -1).norm(dim=1) / (qr.grad.view(B_, -1).norm(dim=1) + 1e-30)
    assert float(e.max()) < tol(dt) * 4, int(e.argmax())
    assert rel(dtab_b, tr.grad) < 2e-2


def test_swin_attention_backward_projection_dgrad_refusals(ops):
    from mvlt_amd._lib import ATTN_SWIN
    for dt, nH in ((torch.float32, 3), (torch.bfloat16, 24), (torch.bfloat16, 4)):
        C_ = 32 * nH
        qkv = rnd((49, 3 * C_), dt, 1)
        table = torch.zeros(169, nH, device="cuda")
        kw = dict(bias_table=table, nW=1, win_res=7, shift=0)
        out, lse = ops.attn_fwd(qkv, ATTN_SWIN, 1, 49, nH, 32, 32 ** -0.5, **kw)
        with pytest.raises(RuntimeError, match="(?i)unsupported"):
            ops.attn_bwd(rnd(out.shape, dt, 2), qkv, out, lse, ATTN_SWIN, 1, 49, nH, 32, 32 ** -0.5, dout_weight=rnd((C_, C_), dt, 3), **kw)


@pytest.mark.parametrize("res,nH,shift,B", [(14, 12, 3, 24), (14, 12, 0, 32), (14, 12, 3, 1), (28, 6, 3, 12), (28, 6, 0, 3), (56, 3, 3, 6), (56, 3, 0, 32)])
def test_swin_wmsa2_backward_one_launch(ops, res, nH, shift, B):
    """mvlt_swin_wmsa2_bwd: projection dgrad + attention backward + qkv dgrad of a Swin block in one launch (unit = two windows x
    three heads, partial qkv-dgrad products per head group) == the three launches, and == the fp32 torch statement of
    visual_feature_extractor.py:224-254 backward (dqkv per window, the SUM of the partial products, the bias-table gradient)."""
    from mvlt_amd._lib import ATTN_SWIN
    dt = torch.bfloat16
    nW = (res // 7) ** 2
    B_, C_ = B * nW, 32 * nH
    assert ops.swin_wmsa2_bwd_parts(dt, B, res, C_, nH) == nH // 3
    qkv = rnd((B_ * 49, 3 * C_), dt, 63)
    table = (0.5 * torch.randn(169, nH, generator=torch.Generator().manual_seed(64))).cuda()
    wproj, wqkv = rnd((C_, C_), dt, 66, C_ ** -0.5), rnd((3 * C_, C_), dt, 67, C_ ** -0.5)
    scale = 32 ** -0.5
    kw = dict(bias_table=table, nW=nW, win_res=res, shift=shift)
    out, lse = ops.attn_fwd(qkv, ATTN_SWIN, B_, 49, nH, 32, scale, **kw)
    dy = rnd(out.shape, dt, 65)
    qr = qkv.float().requires_grad_(True)
    tr = table.clone().requires_grad_(True)
    (swin_ref(qr, tr, nW, res, shift, nH, scale) @ wproj.float().t()).backward(dy.float())
    dtab_a, dtab_b = torch.zeros_like(table), torch.zeros_like(table)
    three = ops.attn_bwd(ops.gemm(dy, wproj, b_kmajor=True), qkv, out, lse, ATTN_SWIN, B_, 49, nH, 32, scale, dbias_table=dtab_a, **kw)
    dxn_three = ops.gemm(three, wqkv, b_kmajor=True)
    one, parts = ops.swin_wmsa2_bwd(dy, qkv, lse, B, res, nH, shift, wproj, wqkv, table, scale, dtab_b)
    torch.cuda.synchronize()
    assert parts.shape == (nH // 3, B_ * 49, C_) and torch.isfinite(one.float()).all() and torch.isfinite(parts.float()).all()
    assert rel(one, three) < 2e-3 and rel(dtab_b, dtab_a) < 2e-3
    assert rel(one, qr.grad) < tol(dt) * 3
    e = (one.float() - qr.grad).view(B_, -1).norm(dim=1) / (qr.grad.view(B_, -1).norm(dim=1) + 1e-30)
    assert float(e.max()) < tol(dt) * 4, int(e.argmax())
    assert rel(dtab_b, tr.grad) < 2e-2
    dxn_ref = qr.grad @ wqkv.float()
    dxn = parts.float().sum(0)
    assert rel(dxn, dxn_ref) < tol(dt) * 3 and rel(dxn, dxn_three) < tol(dt) * 2
    e = (dxn - dxn_ref).view(B_, -1).norm(dim=1) / (dxn_ref.view(B_, -1).norm(dim=1) + 1e-30)
    assert float(e.max()) < tol(dt) * 4, int(e.argmax())


@pytest.mark.parametrize("rows,C_,parts", [(6272, 384, 4), (25088, 192, 2), (3001, 384, 3), (400, 768, 4), (6272, 96, 1)])
def test_layernorm_bwd_sum_of_partial_dy(ops, rows, C_, parts):
    """MvltLayerNormBwd.dy_parts: dy given as 2 .. 4 partial tensors (mvlt_swin_wmsa2_bwd's partial qkv-dgrad products) == the
    plain launch on their bf16-rounded f32 sum, with a gathered dy and a residual-path gradient as the Swin block uses them."""
    dt = torch.bfloat16
    x = rnd((rows, C_), dt, 71)
    gamma = (1.0 + 0.1 * torch.randn(C_, generator=torch.Generator().manual_seed(72))).cuda()
    dyp = rnd((parts, rows, C_), dt, 73, 0.5)
    dres = rnd((rows, C_), dt, 74)
    perm = torch.randperm(rows, generator=torch.Generator().manual_seed(75)).to(torch.int32).cuda()
    mean = x.float().mean(1)
    rstd = (x.float().var(1, unbiased=False) + 1e-5).rsqrt()
    dsum = dyp.float().sum(0).to(dt)
    outs = []
    for dy in (dyp, dsum):
        dg, db = torch.zeros(C_, device="cuda"), torch.zeros(C_, device="cuda")
        dx = ops.layernorm_bwd(dy, x, mean, rstd, gamma, dg, db, dy_rowmap=perm, dres=dres, dy_parts=dy.dim() == 3)
        outs.append((dx, dg, db))
    torch.cuda.synchronize()
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def wmsa_ref(x, w2n, nW, res, shift, nH, g1, b1, wqkv, bqkv, wproj, bproj, table, scale, rowscale):
    """Attention half of SwinTransformerBlock.forward as plain fp32 torch (visual_feature_extractor.py:356-384)."""
    C_ = x.shape[1]
    xn = F.layer_norm(x, (C_,), g1, b1, 1e-5)
    xw = xn[w2n.long()]                                   # roll(-shift) + window_partition
    qkv = xw @ wqkv.t() + bqkv
    ao = swin_ref(qkv, table, nW, res, shift, nH, scale)
    pr = ao @ wproj.t() + bproj
    y = torch.empty_like(pr)
    y[w2n.long()] = pr                                    # window_reverse + roll(+shift)
    if rowscale is not None:
        y = y * rowscale.repeat_interleave(res * res)[:, None]
    return x + y, xw, ao


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("res,C_,shift,B,dp", [(14, 384, 3, 3, True), (14, 384, 0, 2, False), (28, 192, 3, 2, True),
                                              (56, 96, 3, 1, False), (56, 96, 0, 1, True), (14, 512, 3, 1, False),
                                              (28, 256, 0, 1, False), (56, 128, 3, 1, True)])
def test_swin_wmsa_fused_forward(ops, dt, res, C_, shift, B, dp):
    """mvlt_swin_wmsa_fwd (norm1 + shift/partition + qkv + window attention + proj + reverse + DropPath + residual
    in one launch) against the fp32 torch statement, and against the unfused kernel sequence it replaces."""
    from mvlt_amd._lib import ATTN_SWIN
    from mvlt_amd.indexing import batched_window_maps
    nH = C_ // 32
    if not ops.swin_wmsa_supported(dt, C_, nH) or (dt == torch.float32 and C_ == 512):
        pytest.skip("width not covered by the fused kernel in this dtype (LDS)")
    nW = (res // 7) ** 2
    x = rnd((B * res * res, C_), dt, 60)
    g1 = (1.0 + 0.1 * torch.randn(C_, generator=torch.Generator().manual_seed(61))).cuda()
    b1 = (0.1 * torch.randn(C_, generator=torch.Generator().manual_seed(62))).cuda()
    wqkv = rnd((3 * C_, C_), dt, 63, C_ ** -0.5)
    bqkv = (0.1 * torch.randn(3 * C_, generator=torch.Generator().manual_seed(64))).cuda()
    wproj = rnd((C_, C_), dt, 65, C_ ** -0.5)
    bproj = (0.1 * torch.randn(C_, generator=torch.Generator().manual_seed(66))).cuda()
    table = (0.5 * torch.randn(169, nH, generator=torch.Generator().manual_seed(67))).cuda()
    rs = torch.tensor([0.0, 1.25, 1.25][:B] if B > 1 else [1.25]).cuda() if dp else None
    scale = 32 ** -0.5
    w2n, n2w = batched_window_maps(B, res, res, 7, shift, x.device)
    y, (xn, qkv, ao, lse, mean, rstd) = ops.swin_wmsa_fwd(x, w2n, B, res, nH, shift, g1, b1, 1e-5, wqkv, bqkv, wproj, bproj,
                                                      table, scale, rowscale=rs, save=True)
    ref, xw_ref, ao_ref = wmsa_ref(x.float(), w2n, nW, res, shift, nH, g1, b1, wqkv.float(), bqkv, wproj.float(), bproj,
                                   table, scale, rs)
    t = tol(dt)
    assert rel(xn, xw_ref) < t
    assert rel(ao, ao_ref) < t * 2
    assert rel(y, ref) < t * 2
    # eval-mode call (nothing saved) gives the same y
    y2, none = ops.swin_wmsa_fwd(x, w2n, B, res, nH, shift, g1, b1, 1e-5, wqkv, bqkv, wproj, bproj, table, scale, rowscale=rs)
    assert none is None and torch.equal(y2, y)
    # the unfused kernel sequence (LN with row map, qkv GEMM, attention, proj GEMM with scatter + residual)
    xn_u, mean_u, rstd_u, _ = ops.layernorm_fwd(x, g1, b1, 1e-5, out_rowmap=n2w)
    qkv_u = ops.gemm(xn_u, wqkv, bias=bqkv)
    ao_u, lse_u = ops.attn_fwd(qkv_u, ATTN_SWIN, B * nW, 49, nH, 32, scale, bias_table=table, nW=nW, win_res=res, shift=shift)
    y_u = ops.gemm(ao_u, wproj, bias=bproj, residual=x, rowmap=w2n, rowscale=(rs, res * res) if rs is not None else None)
    # same arithmetic, another summation order: a bf16 element may round the other way once in a while
    assert rel(xn, xn_u) < (1e-6 if dt == torch.float32 else 1e-4) and rel(mean, mean_u) < 1e-5 and rel(rstd, rstd_u) < 1e-5
    assert rel(qkv, qkv_u) < t
    assert rel(lse, lse_u) < (1e-5 if dt == torch.float32 else 2e-3)
    assert rel(ao, ao_u) < t and rel(y, y_u) < t
    # head-split mode: one workgroup per (window, head group), projection left to a GEMM launch -- same numbers
    ao_s, (xn_s, qkv_s, ao_s2, lse_s, mean_s, rstd_s) = ops.swin_wmsa_fwd(x, w2n, B, res, nH, shift, g1, b1, 1e-5, wqkv, bqkv, wproj,
                                                                        bproj, table, scale, save=True, head_split=True)
    assert torch.equal(ao_s, ao) and torch.equal(qkv_s, qkv) and torch.equal(xn_s, xn) and torch.equal(lse_s, lse)
    assert torch.equal(mean_s, mean) and torch.equal(rstd_s, rstd)
    y_s = ops.gemm(ao_s, wproj, bias=bproj, residual=x, rowmap=w2n, rowscale=(rs, res * res) if rs is not None else None)
    assert rel(y_s, y) < t


@pytest.mark.parametrize("res,C_,shift,B,dp", [(14, 384, 3, 3, True), (14, 384, 0, 2, False), (14, 384, 3, 32, True),
                                              (28, 192, 3, 2, True), (28, 192, 0, 5, False), (56, 96, 3, 1, False),
                                              (56, 96, 0, 2, True), (14, 384, 5, 1, False),
                                              (14, 384, 3, 48, True), (14, 384, 0, 64, False)])
def test_swin_wmsa2_forward(ops, res, C_, shift, B, dp):
    """mvlt_swin_wmsa2_fwd (two windows per workgroup, head groups across workgroups meeting through attn_out inside the one
    launch) against the fp32 torch statement and the unfused kernel sequence; B = 32 is the stage-2 launch of config #2 (256
    workgroups, every CU waits on three others), B = 3 / 5 leave a partly filled grid; the launch is repeated to show the
    hand-off counters re-arm themselves; B = 48 / 64 at stage 2 run the PERSISTENT multi-round path (384 / 512 units on a grid
    of 256: a workgroup walks several units, re-using its LDS tiles, the Wproj image, the ring prologue and the counters;
    384 is not a multiple of the grid)."""
    from mvlt_amd._lib import ATTN_SWIN
    from mvlt_amd.indexing import batched_window_maps
    dt = torch.bfloat16
    nH = C_ // 32
    assert ops.swin_wmsa2_supported(dt, B, res, C_, nH)
    nW = (res // 7) ** 2
    x = rnd((B * res * res, C_), dt, 60)
    g1 = (1.0 + 0.1 * torch.randn(C_, generator=torch.Generator().manual_seed(61))).cuda()
    b1 = (0.1 * torch.randn(C_, generator=torch.Generator().manual_seed(62))).cuda()
    wqkv = rnd((3 * C_, C_), dt, 63, C_ ** -0.5)
    bqkv = (0.1 * torch.randn(3 * C_, generator=torch.Generator().manual_seed(64))).cuda()
    wproj = rnd((C_, C_), dt, 65, C_ ** -0.5)
    bproj = (0.1 * torch.randn(C_, generator=torch.Generator().manual_seed(66))).cuda()
    table = (0.5 * torch.randn(169, nH, generator=torch.Generator().manual_seed(67))).cuda()
    rs = (0.25 + torch.arange(B, dtype=torch.float32) % 3).cuda() if dp else None
    scale = 32 ** -0.5
    w2n, n2w = batched_window_maps(B, res, res, 7, shift, x.device)
    args = (x, w2n, B, res, nH, shift, g1, b1, 1e-5, wqkv, bqkv, wproj, bproj, table, scale)
    y, (xn, qkv, ao, lse, mean, rstd) = ops.swin_wmsa2_fwd(*args, rowscale=rs, save=True)
    ref, xw_ref, ao_ref = wmsa_ref(x.float(), w2n, nW, res, shift, nH, g1, b1, wqkv.float(), bqkv, wproj.float(), bproj,
                                   table, scale, rs)
    t = tol(dt)
    assert rel(xn, xw_ref) < t
    assert rel(ao, ao_ref) < t * 2
    assert rel(y, ref) < t * 2
    # the unfused kernel sequence
    xn_u, mean_u, rstd_u, _ = ops.layernorm_fwd(x, g1, b1, 1e-5, out_rowmap=n2w)
    qkv_u = ops.gemm(xn_u, wqkv, bias=bqkv)
    ao_u, lse_u = ops.attn_fwd(qkv_u, ATTN_SWIN, B * nW, 49, nH, 32, scale, bias_table=table, nW=nW, win_res=res, shift=shift)
    y_u = ops.gemm(ao_u, wproj, bias=bproj, residual=x, rowmap=w2n, rowscale=(rs, res * res) if rs is not None else None)
    assert rel(xn, xn_u) < 1e-4 and rel(mean, mean_u) < 1e-5 and rel(rstd, rstd_u) < 1e-5
    assert rel(qkv, qkv_u) < t
    assert rel(lse, lse_u) < 2e-3
    assert rel(ao, ao_u) < t and rel(y, y_u) < t
    # every row, not just the norm: a wrong window / head slice would hide in a relative norm at B = 32
    assert float((y.float() - y_u.float()).abs().max()) < 0.25
    # repeated launches (eval mode: nothing saved) reproduce y bit for bit and leave the sync workspace clean
    for _ in range(3):
        y2, none = ops.swin_wmsa2_fwd(*args, rowscale=rs)
        assert none is None and torch.equal(y2, y)
    assert ops.wmsa2_sync_errors() == 0
    ws = ops.wmsa2_sync_ws(x.device, 1)
    assert int(ws.abs().sum().item()) == 0


def _wmsa2_case(ops, B, res=14, C_=384, shift=3, seed=160):
    from mvlt_amd.indexing import batched_window_maps
    dt = torch.bfloat16
    nH = C_ // 32
    x = rnd((B * res * res, C_), dt, seed)
    g1 = (1.0 + 0.1 * torch.randn(C_, generator=torch.Generator().manual_seed(seed + 1))).cuda()
    b1 = (0.1 * torch.randn(C_, generator=torch.Generator().manual_seed(seed + 2))).cuda()
    wqkv = rnd((3 * C_, C_), dt, seed + 3, C_ ** -0.5)
    bqkv = (0.1 * torch.randn(3 * C_, generator=torch.Generator().manual_seed(seed + 4))).cuda()
    wproj = rnd((C_, C_), dt, seed + 5, C_ ** -0.5)
    bproj = (0.1 * torch.randn(C_, generator=torch.Generator().manual_seed(seed + 6))).cuda()
    table = (0.5 * torch.randn(169, nH, generator=torch.Generator().manual_seed(seed + 7))).cuda()
    w2n, _ = batched_window_maps(B, res, res, 7, shift, x.device)
    return (x, w2n, B, res, nH, shift, g1, b1, 1e-5, wqkv, bqkv, wproj, bproj, table, 32 ** -0.5), w2n


def test_swin_wmsa2_beside_a_kernel_that_holds_cus(ops):
    """The hand-off of mvlt_swin_wmsa2_fwd needs the four head-group workgroups of a window pair resident together.  Here other
    streams hold CUs while the stage-2 launch of config #2 (256 workgroups, one per CU) runs -- what the single-workgroup plan
    kernels on the side stream do in the training step, made much worse: (a) one long single-workgroup kernel with a large
    LDS footprint, (b) 64 of them.  The launch only gets the CUs late; the result must be bit-identical to the undisturbed
    launch and no wait may run out."""
    args, _ = _wmsa2_case(ops, 32)
    y0, _ = ops.swin_wmsa2_fwd(*args)
    torch.cuda.synchronize()
    other = torch.cuda.Stream()
    for blocks in (1, 64):
        ops.debug_hold_cus(blocks, 100 * 1024, 3000, stream=other)          # 3 ms, 100 KB of LDS each: no wmsa2 workgroup fits beside one
        for _ in range(4):
            y, _ = ops.swin_wmsa2_fwd(*args)
            assert torch.equal(y, y0)
        torch.cuda.synchronize()
    assert ops.wmsa2_sync_errors() == 0
    ops.wmsa2_check(sync=True)
    ws = ops.wmsa2_sync_ws(args[0].device, 1)
    assert int(ws.abs().sum().item()) == 0


def test_swin_wmsa2_timeout_is_loud(ops):
    """A hand-off wait that runs out must not pass silently: the unit's rows of y are NaN, the sticky
    error count (word 0 of the workspace) is raised, ops.wmsa2_check raises both in its synchronous and in its one-call-late
    asynchronous form, and after wmsa2_clear_errors the next launch is clean again.  Provoked by mis-arming the arrival
    flags of ONE window pair (so its four groups never see each other) with the wait shortened to 20 ms."""
    args, w2n = _wmsa2_case(ops, 8)
    x = args[0]
    y0, _ = ops.swin_wmsa2_fwd(*args)
    assert ops.wmsa2_sync_errors() == 0
    ws = ops.wmsa2_sync_ws(x.device, 1)
    bad_set = 5
    try:
        ops.wmsa2_set_timeout_ms(20)
        ws[16 + 8 * bad_set:16 + 8 * bad_set + 4] = -1000   # the four arrival flags of pair `bad_set` can never read as raised
        y, _ = ops.swin_wmsa2_fwd(*args)
        torch.cuda.synchronize()
        assert ops.wmsa2_sync_errors() == 4               # the four head groups of the pair
        rows = w2n[bad_set * 98:(bad_set + 1) * 98].long()          # token rows of the pair's two windows
        assert bool(torch.isnan(y[rows].float()).all())
        keep = torch.ones(y.shape[0], dtype=torch.bool, device=y.device)
        keep[rows] = False
        assert torch.equal(y[keep], y0[keep])             # every other pair is untouched
        with pytest.raises(ops.DeviceHandoffError):
            ops.wmsa2_check(sync=True)
        ops.wmsa2_check(sync=False)                       # queues the copy of the error count ...
        torch.cuda.synchronize()
        with pytest.raises(ops.DeviceHandoffError):
            ops.wmsa2_check(sync=False)                   # ... and the next call sees it, without a device sync of its own
    finally:
        ops.wmsa2_set_timeout_ms(0)
        ops.wmsa2_clear_errors()
    y2, _ = ops.swin_wmsa2_fwd(*args)
    assert torch.equal(y2, y0) and ops.wmsa2_sync_errors() == 0
    ops.wmsa2_check(sync=True)


def bert_ref(qkv, B, Lq, nH, mask_add, scale, keep=None, p=0.0):
    hd = qkv.shape[1] // (3 * nH)
    q, k, v = qkv.view(B, Lq, 3, nH, hd).permute(2, 0, 3, 1, 4)
    att = (q @ k.transpose(-1, -2)) * scale + mask_add
    pr = att.softmax(-1)
    if keep is not None:
        pr = pr * keep / (1 - p)
    return (pr @ v).transpose(1, 2).reshape(B * Lq, nH * hd)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("T,seq2seq,p", [(80, False, 0.0), (80, True, 0.0), (23, False, 0.0), (24, True, 0.1),
                                         (80, False, 0.1), (128, False, 0.1), (128, True, 0.0),   # 128: config #5 (L=179)
                                         (109, True, 0.1), (109, False, 0.0), (110, False, 0.1),   # L=160: the longest sequence of the 5-wave one-launch bf16 backward (10 key tiles); 161: the 6-wave form (round 5: up to 192 rows, config #5's 179)
                                         (141, True, 0.1), (141, False, 0.0)])   # L=192: its longest
def test_bert_attention(ops, dt, T, seq2seq, p):
    from mvlt_amd._lib import ATTN_BIDIR, ATTN_SEQ2SEQ
    from oracle import mvlt_oracle as O
    B, nH, n_img = 3, 4, 49
    Lq = n_img + 2 + T
    qkv = rnd((B * Lq, 3 * nH * 64), dt, 50)
    ids = torch.zeros(B, T, dtype=torch.long)
    for b, ln in enumerate((T, max(1, T // 3), 1)):
        ids[b, :ln] = 5 + torch.arange(ln)
    if seq2seq:
        mask = O.additive_mask(O.seq2seq_bool_mask(Lq, n_img + 1)[None].expand(B, Lq, Lq)).cuda()
        mode = ATTN_SEQ2SEQ
    else:
        mask = O.additive_mask(O.bidir_bool_mask(ids, B, n_img)).cuda()
        mode = ATTN_BIDIR
    kw = dict(text_ids=ids.cuda(), obj_end=n_img + 1, dropout=(p, 99, 5))
    out, lse = ops.attn_fwd(qkv, mode, B, Lq, nH, 64, 0.125, **kw)
    keep = None
    if p > 0:
        keep = ops.dropout_mask(B * nH * Lq * Lq, p, 99, 5, qkv.device).view(B, nH, Lq, Lq).float()
    qr = qkv.float().requires_grad_(True)
    ref = bert_ref(qr, B, Lq, nH, mask, 0.125, keep, p)
    assert rel(out, ref) < tol(dt)
    dout = rnd(out.shape, dt, 51)
    ref.backward(dout.float())
    dqkv = ops.attn_bwd(dout, qkv, out, lse, mode, B, Lq, nH, 64, 0.125, **kw)
    assert rel(dqkv, qr.grad) < tol(dt) * 3


@pytest.mark.parametrize("case", ["swin_bf16", "swin_f32", "bert_bf16_131", "bert_f32_131", "bert_bf16_179"])
def test_attention_backward_with_stop_event(ops, case):
    """mvlt_attn_bwd_ev: same gradient as mvlt_attn_bwd on every backward path (one-launch bf16 kernels, the generic f32
    kernel, the two-launch form of long rows), and a second stream that waits for the caller's event -- bound to the call's
    last kernel, no marker packet -- sees the complete gradient."""
    import ctypes
    from mvlt_amd._lib import ATTN_SWIN, ATTN_BIDIR
    hip = ctypes.CDLL("libamdhip64.so")
    ev = ctypes.c_void_p()
    assert hip.hipEventCreateWithFlags(ctypes.byref(ev), ctypes.c_uint(0x2)) == 0          # hipEventDisableTiming
    dt = torch.float32 if "f32" in case else torch.bfloat16
    if case.startswith("swin"):
        res, nH, B = 14, 12, 32
        nW = (res // 7) ** 2
        qkv = rnd((B * nW * 49, 3 * nH * 32), dt, 60)
        table = (0.5 * torch.randn(169, nH, generator=torch.Generator().manual_seed(61))).cuda()
        args = (ATTN_SWIN, B * nW, 49, nH, 32, 32 ** -0.5)
        kw = dict(bias_table=table, nW=nW, win_res=res, shift=3)
        bkw = lambda: dict(kw, dbias_table=torch.zeros_like(table))
    else:
        T = 128 if case.endswith("179") else 80
        B, nH, n_img = 8, 12, 49
        Lq = n_img + 2 + T
        qkv = rnd((B * Lq, 3 * nH * 64), dt, 62)
        ids = torch.zeros(B, T, dtype=torch.long)
        for b in range(B):
            ids[b, :(b * 37) % (T + 1)] = 7
        args = (ATTN_BIDIR, B, Lq, nH, 64, 0.125)
        kw = dict(text_ids=ids.cuda(), obj_end=n_img + 1, dropout=(0.1, 5, 3))
        bkw = lambda: dict(kw)
    out, lse = ops.attn_fwd(qkv, *args, **kw)
    dout = rnd(out.shape, dt, 63)
    want = ops.attn_bwd(dout, qkv, out, lse, *args, **bkw())
    side = torch.cuda.Stream()
    for _ in range(3):
        got = ops.attn_bwd(dout, qkv, out, lse, *args, event=ev.value, **bkw())
        assert hip.hipStreamWaitEvent(ctypes.c_void_p(side.cuda_stream), ev, 0) == 0
        with torch.cuda.stream(side):
            seen = got.clone()                     # ordered behind the event only
        side.synchronize()
        assert hip.hipEventQuery(ev) == 0
        torch.cuda.synchronize()
        assert torch.equal(seen, want) and torch.equal(got, want)
        got.record_stream(side)
    assert hip.hipEventDestroy(ev) == 0


@pytest.mark.parametrize("seq2seq", [False, True])
def test_bert_attention_backward_full_batch_per_sequence(ops, seq2seq):
    """bf16 MVLBert attention backward at the step's size (B=32, 12 heads, L=131, dropout 0.1, every caption length
    from empty to full): the one-launch scores-once kernel against the fp32 torch statement of modeling_bert.py:111-136,
    globally and per sequence (a workgroup that mixed up two sequences or mis-sized its last key tile would hide in the
    global norm)."""
    from mvlt_amd._lib import ATTN_BIDIR, ATTN_SEQ2SEQ
    from oracle import mvlt_oracle as O
    dt = torch.bfloat16
    B, nH, n_img, T, p = 32, 12, 49, 80, 0.1
    Lq = n_img + 2 + T
    qkv = rnd((B * Lq, 3 * nH * 64), dt, 52)
    ids = torch.zeros(B, T, dtype=torch.long)
    for b in range(B):
        ln = (b * 83) % (T + 1)                     # 0 .. 80, including 0, 77, 78, 79 (9 key tiles) and 80
        ids[b, :ln] = 5 + torch.arange(ln)
    if seq2seq:
        mask = O.additive_mask(O.seq2seq_bool_mask(Lq, n_img + 1)[None].expand(B, Lq, Lq)).cuda()
        mode = ATTN_SEQ2SEQ
    else:
        mask = O.additive_mask(O.bidir_bool_mask(ids, B, n_img)).cuda()
        mode = ATTN_BIDIR
    kw = dict(text_ids=ids.cuda(), obj_end=n_img + 1, dropout=(p, 77, 6))
    out, lse = ops.attn_fwd(qkv, mode, B, Lq, nH, 64, 0.125, **kw)
    keep = ops.dropout_mask(B * nH * Lq * Lq, p, 77, 6, qkv.device).view(B, nH, Lq, Lq).float()
    qr = qkv.float().requires_grad_(True)
    ref = bert_ref(qr, B, Lq, nH, mask, 0.125, keep, p)
    dout = rnd(out.shape, dt, 53)
    ref.backward(dout.float())
    dqkv = ops.attn_bwd(dout, qkv, out, lse, mode, B, Lq, nH, 64, 0.125, **kw)
    assert rel(dqkv, qr.grad) < tol(dt) * 3
    e = (dqkv.float() - qr.grad).view(B, -1).norm(dim=1) / (qr.grad.view(B, -1).norm(dim=1) + 1e-30)
    assert float(e.max()) < tol(dt) * 4, int(e.argmax())


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("seq2seq,p", [(False, 0.0), (True, 0.0), (False, 0.1)])
@pytest.mark.parametrize("T,lens_t", [(40, [40, 13, 1, 27]), (80, [80, 79, 78, 77, 0, 13, 45, 61])])
def test_bert_attention_packed_rows_equal_dense(ops, dt, seq2seq, p, T, lens_t):
    """MvltAttn.row_start / seq_len: the kept rows of a packed launch == the same rows of the dense launch
    (forward, backward, same dropout mask -- its index is (sequence, head, q, k) in both layouts).  T = 80: the
    step's sequence length, with captions that end in the ninth key tile (78..80), just before it (77), and an
    empty one."""
    from mvlt_amd._lib import ATTN_BIDIR, ATTN_SEQ2SEQ
    B, nH, n_img = len(lens_t), 4, 49
    Lq = n_img + 2 + T
    ids = torch.zeros(B, T, dtype=torch.long)
    for b, ln in enumerate(lens_t):
        ids[b, :ln] = 5 + torch.arange(ln)
    seq_len = torch.tensor([n_img + 2 + ln for ln in lens_t], dtype=torch.int32)
    row_start = (torch.cumsum(seq_len, 0) - seq_len).to(torch.int32)
    R = int(seq_len.sum())
    dense_rows = torch.cat([b * Lq + torch.arange(int(seq_len[b])) for b in range(B)]).cuda()
    qkv = rnd((B * Lq, 3 * nH * 64), dt, 70)
    dout = rnd((B * Lq, nH * 64), dt, 71)
    mode = ATTN_SEQ2SEQ if seq2seq else ATTN_BIDIR
    kw = dict(text_ids=ids.cuda(), obj_end=n_img + 1, dropout=(p, 99, 5))
    out_d, lse_d = ops.attn_fwd(qkv, mode, B, Lq, nH, 64, 0.125, **kw)
    # the dense backward with zero upstream gradient on the dropped rows == what the loss gives them
    dout_d = torch.zeros_like(dout)
    dout_d[dense_rows] = dout[dense_rows]
    dqkv_d = ops.attn_bwd(dout_d, qkv, out_d, lse_d, mode, B, Lq, nH, 64, 0.125, **kw)
    pack = (row_start.cuda(), seq_len.cuda(), R)
    qkv_p, dout_p = qkv[dense_rows].contiguous(), dout[dense_rows].contiguous()
    out_p, lse_p = ops.attn_fwd(qkv_p, mode, B, Lq, nH, 64, 0.125, pack=pack, **kw)
    assert out_p.shape == (R, nH * 64)
    assert rel(out_p, out_d[dense_rows]) < 1e-6 if dt == torch.float32 else rel(out_p, out_d[dense_rows]) < 1e-3
    dqkv_p = ops.attn_bwd(dout_p, qkv_p, out_p, lse_p, mode, B, Lq, nH, 64, 0.125, pack=pack, **kw)
    assert rel(dqkv_p, dqkv_d[dense_rows]) < (1e-5 if dt == torch.float32 else 5e-3)


@pytest.mark.parametrize("dt", DT)
def test_embed_packed_rows_and_zero_batch(ops, dt):
    B, n_img, T, H = 3, 49, 24, 256
    g = torch.Generator().manual_seed(80)
    word, pos, typ = (torch.randn(n, H, generator=g).cuda() for n in (3001, 512, 3))
    feat = rnd((B, n_img, H), dt, 81)
    lens_t = [24, 7, 1]
    ids = torch.zeros(B, T, dtype=torch.long)
    for b, ln in enumerate(lens_t):
        ids[b, :ln] = torch.randint(5, 3000, (ln,), generator=g)
    ids = ids.cuda()
    seq_len = torch.tensor([n_img + 2 + ln for ln in lens_t], dtype=torch.int32)
    row_start = (torch.cumsum(seq_len, 0) - seq_len).to(torch.int32)
    R = int(seq_len.sum())
    Lq = n_img + 2 + T
    rows = torch.cat([b * Lq + torch.arange(int(seq_len[b])) for b in range(B)]).cuda()
    pack = (row_start.cuda(), seq_len.cuda(), R)
    dense = ops.embed_fwd(ids, feat, word, pos, typ, 101, 102).view(B * Lq, H)
    packed = ops.embed_fwd(ids, feat, word, pos, typ, 101, 102, pack=pack)
    assert packed.shape == (R, H) and torch.equal(packed, dense[rows])
    dout = rnd((B * Lq, H), dt, 82)
    dout_zeroed = torch.zeros_like(dout)
    dout_zeroed[rows] = dout[rows]
    gd = [torch.zeros_like(t) for t in (word, pos, typ)]
    gp = [torch.zeros_like(t) for t in (word, pos, typ)]
    dimg_d = ops.embed_bwd(dout_zeroed.view(B, Lq, H), ids, n_img, word, pos, typ, 101, 102, *gd)
    dimg_p = ops.embed_bwd(dout[rows].contiguous(), ids, n_img, word, pos, typ, 101, 102, *gp, pack=pack, B=B)
    assert torch.equal(dimg_p, dimg_d)
    for a, b in zip(gp, gd):
        # the dense run also adds the (zero) gradients of the padded positions to word id 0: same sums
        assert rel(a, b) < 1e-5
    bufs = [torch.randn(n, device="cuda") for n in (169 * 3, 169 * 24, 5, 4096)]
    ops.zero_batch(bufs)
    assert all(float(t.abs().max()) == 0.0 for t in bufs)


# ------------------------------------------------------------------ data movement, loss, optimizer
@pytest.mark.parametrize("dt", DT)
def test_im2col_and_embed(ops, dt):
    img = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(60)).cuda()
    cols = ops.im2col_patch(img, dt, 4)
    w = torch.randn(96, 3, 4, 4, generator=torch.Generator().manual_seed(61)).cuda()
    ref = F.conv2d(img, w, stride=4).flatten(2).transpose(1, 2).reshape(-1, 96)
    assert rel(cols.float() @ w.view(96, -1).t(), ref) < tol(dt)
    # embeddings
    B, n_img, T, H = 3, 49, 24, 256
    word = torch.randn(3001, H, generator=torch.Generator().manual_seed(62)).cuda()
    pos = torch.randn(512, H, generator=torch.Generator().manual_seed(63)).cuda()
    typ = torch.randn(3, H, generator=torch.Generator().manual_seed(64)).cuda()
    ids = torch.randint(0, 3000, (B, T), generator=torch.Generator().manual_seed(65)).cuda()
    ids[0, 5:] = 0
    feat = rnd((B, n_img, H), dt, 66)
    out = ops.embed_fwd(ids, feat, word, pos, typ, 101, 102)
    wr, pr, tr = word.clone().requires_grad_(True), pos.clone().requires_grad_(True), typ.clone().requires_grad_(True)
    fr = feat.float().requires_grad_(True)
    Lq = n_img + 2 + T
    src = torch.cat([wr[101].expand(B, 1, H), fr, wr[102].expand(B, 1, H), wr[ids]], 1)
    tt = (torch.arange(Lq, device="cuda") <= n_img + 1).long()
    ref = src + tr[tt][None] + pr[:Lq][None]
    assert rel(out, ref) < tol(dt)
    dout = rnd((B, Lq, H), dt, 67)
    ref.backward(dout.float())
    # the word table is a scatter-add target (cleared by the caller); the position / type tables are OVERWRITTEN whole
    # (ordered batch sums, unused rows zeroed): they start from garbage here
    dw, dp_, dt_ = torch.zeros_like(word), torch.full_like(pos, 7.0), torch.full_like(typ, -3.0)
    dimg = ops.embed_bwd(dout, ids, n_img, word, pos, typ, 101, 102, dw, dp_, dt_)
    assert rel(dimg, fr.grad) < tol(dt)
    assert rel(dw, wr.grad) < 1e-4 and rel(dp_, pr.grad) < 1e-4 and rel(dt_, tr.grad) < 1e-4
    assert float(dp_[Lq:].abs().max()) == 0.0 and float(dt_[2].abs().max()) == 0.0
    dp2, dt2 = torch.full_like(pos, 1.0), torch.full_like(typ, 1.0)
    ops.embed_bwd(dout, ids, n_img, word, pos, typ, 101, 102, torch.zeros_like(word), dp2, dt2)
    assert torch.equal(dp2, dp_) and torch.equal(dt2, dt_)          # bit-reproducible


@pytest.mark.parametrize("dt", DT)
def test_rows_transform_cast_unary(ops, dt):
    x = rnd((120, 64), dt, 70)
    perm = torch.randperm(120, generator=torch.Generator().manual_seed(3)).int().cuda()
    rs = torch.tensor([0.0, 2.0, 1.5], device="cuda")
    out = ops.rows_transform(x, rowmap=perm, rowscale=(rs, 40))
    exp = x.float()[perm.long()] * rs[perm.long() // 40][:, None]
    assert rel(out, exp) < tol(dt)
    out = ops.rows_transform(x, dropout=(0.3, 5, 6))
    keep = ops.dropout_mask(120 * 64, 0.3, 5, 6, x.device).view(120, 64).float()
    assert rel(out, x.float() * keep / 0.7) < tol(dt)
    assert rel(ops.gelu(x), F.gelu(x.float())) < tol(dt)
    y = ops.tanh_fwd(x)
    assert rel(y, torch.tanh(x.float())) < tol(dt)
    assert rel(ops.tanh_bwd(y, x), x.float() * (1 - y.float() ** 2)) < tol(dt)
    z = torch.randn(1003, generator=torch.Generator().manual_seed(71)).cuda()
    zb = ops.cast(z, torch.bfloat16)
    assert torch.equal(zb, z.to(torch.bfloat16))
    assert torch.equal(ops.cast(zb, torch.float32), zb.float())
    s = ops.droppath_scale(1000, 0.3, 11, 12, z.device)
    assert all(v == 0.0 or abs(v - 1 / 0.7) < 1e-6 for v in s.unique().tolist())
    assert 0.6 < (s > 0).float().mean().item() < 0.8
    probs = torch.tensor([0.0, 0.1, 0.5, 0.3], device="cuda")
    sc = ops.droppath_scales(probs, 4000, 21, 5)                     # every DropPath row of a forward pass in one launch
    assert sc.shape == (4, 4000) and bool((sc[0] == 1.0).all())
    for r, p in enumerate(probs.tolist()):
        assert all(v == 0.0 or abs(v - 1 / (1 - p)) < 1e-6 for v in sc[r].unique().tolist())
        assert abs((sc[r] > 0).float().mean().item() - (1 - p)) < 0.04
    assert torch.equal(sc, ops.droppath_scales(probs, 4000, 21, 5)) and not torch.equal(sc, ops.droppath_scales(probs, 4000, 22, 5))
    assert torch.equal(sc[3], ops.droppath_scale(4000, 0.3, 21, 5 + 3, probs.device))      # row r == the one-row form with tag + r


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("V,ld", [(3000, 3008), (30522, 30528), (2, 4)])
def test_cross_entropy(ops, dt, V, ld):
    rows = 37
    logits = torch.zeros(rows, ld, dtype=dt, device="cuda")
    logits[:, :V] = rnd((rows, V), dt, 80, 3.0)
    labels = torch.randint(0, V, (rows,), generator=torch.Generator().manual_seed(81))
    labels[::3] = -100
    labels = labels.cuda()
    acc, lse = ops.ce_fwd(logits, V, labels)
    lr = logits[:, :V].float().requires_grad_(True)
    ref = F.cross_entropy(lr, labels, ignore_index=-100)
    assert abs(acc[0].item() / acc[1].item() - ref.item()) < 1e-4 * abs(ref.item())
    assert acc[1].item() == (labels >= 0).sum().item()
    ref.backward()
    d = ops.ce_bwd(logits.clone(), V, labels, lse, acc)
    assert rel(d[:, :V], lr.grad) < (1e-5 if dt == torch.float32 else 1e-2)
    assert float(d[:, V:].float().abs().sum()) == 0.0


def test_adamw_matches_torch(ops):
    n = 10007
    p0 = torch.randn(n, generator=torch.Generator().manual_seed(90))
    pt = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pt], lr=4e-5, betas=(0.9, 0.999), eps=1e-6, weight_decay=1e-4)
    p = torch.zeros(n + 1, device="cuda")[:n].copy_(p0)          # 16B-aligned base
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    sh = torch.zeros(n, dtype=torch.bfloat16, device="cuda")
    for step in range(1, 4):
        g = torch.randn(n, generator=torch.Generator().manual_seed(90 + step))
        pt.grad = g.clone()
        opt.step()
        ops.adamw(p, g.cuda(), m, v, sh, 4e-5, 0.9, 0.999, 1e-6, 1e-4, step)
    assert rel(p, pt.detach()) < 1e-6
    assert torch.equal(sh, p.to(torch.bfloat16))


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("past,n_new,cap,dev_past", [(51, 2, 80, False), (200, 2, 202, True), (3, 1, 16, False),
                                                     (100, 4, 128, False), (97, 2, 600, False), (60, 5, 80, True)])
def test_cached_attention_and_argmax(ops, dt, past, n_new, cap, dev_past):
    """mvlt_attn_cached (one wave per (b, head); serial fallback for > 512 cache slots or > 4 new rows)."""
    B, nH, hd = 2, 4, 64
    kc = torch.zeros(B, nH, cap, hd, dtype=dt, device="cuda")
    vc = torch.zeros_like(kc)
    kc[:, :, :past] = rnd((B, nH, past, hd), dt, 100)
    vc[:, :, :past] = rnd((B, nH, past, hd), dt, 101)
    qkv = rnd((B * n_new, 3 * nH * hd), dt, 102)
    kref, vref = kc.clone(), vc.clone()
    out = ops.attn_cached(qkv, kc, vc, torch.tensor([past], dtype=torch.int32, device="cuda") if dev_past else past, 0.125)
    q, k, v = qkv.float().view(B, n_new, 3, nH, hd).permute(2, 0, 3, 1, 4)
    K = torch.cat([kref[:, :, :past].float(), k], 2)
    V_ = torch.cat([vref[:, :, :past].float(), v], 2)
    att = (q @ K.transpose(-1, -2)) * 0.125
    causal = torch.arange(past + n_new, device="cuda")[None, :] <= past + torch.arange(n_new, device="cuda")[:, None]
    att = att.masked_fill(~causal[None, None], float("-inf")).softmax(-1)
    ref = (att @ V_).transpose(1, 2).reshape(B * n_new, nH * hd)
    assert rel(out, ref) < tol(dt)
    assert rel(kc[:, :, past:past + n_new], k) < 1e-6 and rel(vc[:, :, past:past + n_new], v) < 1e-6
    logits = rnd((5, 3008), dt, 103)
    assert torch.equal(ops.argmax(logits, 3000), logits[:, :3000].float().argmax(-1))


@pytest.mark.parametrize("dt", DT)
def test_layernorm_bwd_branch_output(ops, dt):
    """Second output of LN backward: dz[map[r]] = dropmask(dx[r]) * rowscale[r // rps]
    (Swin window scatter + DropPath, BERT hidden-dropout backward)."""
    rows, C = 300, 192
    x, dy = rnd((rows, C), dt, 120, 2.0), rnd((rows, C), dt, 121)
    g = (1 + 0.1 * torch.randn(C, generator=torch.Generator().manual_seed(122))).cuda()
    b = torch.zeros(C, device="cuda")
    _, mean, rstd, _ = ops.layernorm_fwd(x, g, b, 1e-5)
    dgam, dbet = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ref_dx = ops.layernorm_bwd(dy, x, mean, rstd, g, dgam, dbet)
    perm = torch.randperm(rows, generator=torch.Generator().manual_seed(4)).int().cuda()
    rs = torch.tensor([0.0, 1.5, 2.0], device="cuda")
    dx, dz = ops.layernorm_bwd(dy, x, mean, rstd, g, dgam, dbet, branch=dict(rowmap=perm, rowscale=(rs, 100)))
    assert torch.equal(dx, ref_dx)
    exp = torch.empty_like(ref_dx, dtype=torch.float32)
    exp[perm.long()] = ref_dx.float() * rs[torch.arange(rows, device="cuda") // 100][:, None]
    assert rel(dz, exp) < tol(dt)
    dx, dz = ops.layernorm_bwd(dy, x, mean, rstd, g, dgam, dbet, branch=dict(dropout=(0.2, 9, 3)))
    keep = ops.dropout_mask(rows * C, 0.2, 9, 3, x.device).view(rows, C).float()
    assert rel(dz, ref_dx.float() * keep / 0.8) < tol(dt)


def test_gemm_plan_introspection(ops):
    import ctypes as C
    from mvlt_amd import _lib as L
    p = L.MvltGemm()
    p.dtype, p.M, p.N, p.K = L.BF16, 3072, 768, 4192          # BERT FFN-in weight gradient
    p.a_kmajor = p.b_kmajor = 1
    bm, bn, sp = C.c_int(), C.c_int(), C.c_int()
    assert L.lib().mvlt_gemm_plan(C.byref(p), C.byref(bm), C.byref(bn), C.byref(sp)) == 0
    assert (bm.value, bn.value) in ((64, 128), (128, 128)) and sp.value >= 1


# ------------------------------------------------------------------ ragged batches planned on the device
def test_pack_plan_matches_host_statement(ops):
    """mvlt_pack_plan against the plain statement of its contract (INT: bit-exact)."""
    torch.manual_seed(5)
    B, T, n_img = 7, 19, 49
    ids = torch.randint(1, 50, (B, T))
    lab = torch.full((B, T), -100)
    for b, ln in enumerate((0, 19, 5, 1, 18, 7, 3)):
        ids[b, ln:] = 0
    ids[2, 1] = 0
    lab[5, 9] = 3                                   # label beyond the last non-zero id
    rs, sl, tot, rs64, trow = ops.pack_plan(ids.cuda(), lab.cuda(), n_img)
    keep = (ids != 0) | (lab >= 0)
    lens = torch.tensor([int(keep[b].nonzero().max()) + 1 if keep[b].any() else 0 for b in range(B)])
    want_sl = lens + n_img + 2
    want_rs = torch.cumsum(want_sl, 0) - want_sl
    assert torch.equal(sl.cpu().long(), want_sl) and torch.equal(rs.cpu().long(), want_rs)
    assert int(tot.item()) == int(want_sl.sum()) and torch.equal(rs64.cpu(), want_rs)
    t = torch.arange(T)[None, :]
    want_row = torch.where(t < lens[:, None], want_rs[:, None] + n_img + 2 + t, want_rs[:, None].expand(B, T))
    assert torch.equal(trow.cpu().view(B, T), want_row)


@pytest.mark.parametrize("dt", DT)
def test_device_row_count_gemm_and_layernorm(ops, dt):
    """MvltGemm.m_dev / MvltLayerNorm.rows_dev: the launch is sized for the upper bound, the kernels read the real row
    count from device memory.  Rows beyond it are poisoned with NaN (never read) and the outputs beyond it keep their
    sentinel (never written)."""
    Mx, N, K, R = 333, 192, 256, 200
    rd = torch.tensor([R], dtype=torch.int32).cuda()
    a = rnd((Mx, K), dt, 1); w = rnd((N, K), dt, 2, K ** -0.5); bias = torch.randn(N).cuda()
    a[R:] = float("nan")
    out = torch.full((Mx, N), 7.0, dtype=dt, device="cuda")
    ops.gemm(a, w, bias=bias, out=out, m_dev=rd)
    ref = a[:R].float() @ w.float().t() + bias
    assert rel(out[:R], ref) < tol(dt) and bool((out[R:] == 7.0).all())
    # dgrad layout (B k-major)
    wk = rnd((K, N), dt, 3, K ** -0.5)
    out.fill_(7.0)
    ops.gemm(a, wk, b_kmajor=True, out=out, m_dev=rd)
    assert rel(out[:R], a[:R].float() @ wk.float()) < tol(dt) and bool((out[R:] == 7.0).all())
    # weight gradient: the reduction stops at the device count (single product and grouped launch)
    dy = rnd((Mx, N), dt, 4); dy[R:] = float("nan")
    dw = torch.empty((N, K), device="cuda"); db = torch.empty(N, device="cuda")
    ops.gemm(dy, a, a_kmajor=True, b_kmajor=True, out=dw, out_f32=True, a_colsum=db, m_dev=rd)
    assert rel(dw, dy[:R].float().t() @ a[:R].float()) < tol(dt) and rel(db, dy[:R].float().sum(0)) < tol(dt)
    big_dy = rnd((Mx, 768), dt, 5); big_dy[R:] = float("nan")
    big_x = rnd((Mx, 768), dt, 6); big_x[R:] = float("nan")
    items = [(big_dy, big_x, torch.empty((768, 768), device="cuda"), torch.empty(768, device="cuda"), rd) for _ in range(3)]
    ops.wgrad_group(items)
    for it in items:
        assert rel(it[2], big_dy[:R].float().t() @ big_x[:R].float()) < tol(dt)
        assert rel(it[3], big_dy[:R].float().sum(0)) < tol(dt)
    # LayerNorm forward / backward
    C_ = 256
    x = rnd((Mx, C_), dt, 7); x[R:] = float("nan")
    g = (1.0 + 0.1 * torch.randn(C_)).cuda(); b = (0.1 * torch.randn(C_)).cuda()
    y = torch.full((Mx, C_), 7.0, dtype=dt, device="cuda")
    _, mean, rstd, _ = ops.layernorm_fwd(x, g, b, 1e-5, out=y, rows_dev=rd)
    yr = torch.nn.functional.layer_norm(x[:R].float(), (C_,), g, b, 1e-5)
    assert rel(y[:R], yr) < tol(dt) and bool((y[R:] == 7.0).all())
    dyl = rnd((Mx, C_), dt, 8); dyl[R:] = float("nan")
    dg, dbt = torch.empty(C_, device="cuda"), torch.empty(C_, device="cuda")
    dx = torch.full((Mx, C_), 7.0, dtype=dt, device="cuda")
    ops.layernorm_bwd(dyl, x, mean, rstd, g, dg, dbt, dx=dx, rows_dev=rd)
    xr = x[:R].float().requires_grad_(True)
    gr = g.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr, (C_,), gr, br, 1e-5).backward(dyl[:R].float())
    t = tol(dt) * 2
    assert rel(dx[:R], xr.grad) < t and rel(dg, gr.grad) < t and rel(dbt, br.grad) < t and bool((dx[R:] == 7.0).all())


def test_label_plan_and_rows_scatter(ops):
    """mvlt_label_plan (labelled positions first, stable; INT: bit-exact) and mvlt_rows_scatter (backward of the gather)."""
    torch.manual_seed(9)
    N, H = 37 * 24, 64
    labels = torch.full((N,), -100, dtype=torch.int64)
    pick = torch.randperm(N)[:101]
    labels[pick] = torch.randint(0, 3000, (101,))
    text_row = torch.randperm(4 * N)[:N].to(torch.int64)                 # distinct packed rows
    gr, sel, cnt = ops.label_plan(labels.cuda(), text_row.cuda())
    order = torch.cat([torch.nonzero(labels >= 0).flatten(), torch.nonzero(labels < 0).flatten()])
    assert int(cnt.item()) == 101
    assert torch.equal(gr.cpu().long(), text_row[order]) and torch.equal(sel.cpu(), labels[order])
    gr2, sel2, cnt2 = ops.label_plan(labels.cuda(), None)                # identity rows
    assert torch.equal(gr2.cpu().long(), order) and int(cnt2.item()) == 101
    for dt in DT:
        dx = rnd((N, H), dt, 10)
        out = torch.zeros((4 * N, H), dtype=dt, device="cuda")
        ops.rows_scatter(dx, gr, cnt, out)
        ref = torch.zeros((4 * N, H), dtype=dt)
        ref[text_row[order][:101]] = dx.cpu()[:101]
        assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("dt", DT)
def test_ragged_cross_entropy(ops, dt):
    """mvlt_ce_fwd_ragged / _bwd_ragged: rows beyond the device-side count are neither read (NaN poison) nor written."""
    rows, V, R = 40, 1000, 23
    ld = 1024
    logits = rnd((rows, ld), dt, 11)
    logits[R:] = float("nan")
    labels = torch.randint(0, V, (rows,))
    labels[3] = -100
    rd = torch.tensor([R], dtype=torch.int32).cuda()
    acc, lse = ops.ce_fwd(logits, V, labels.cuda(), rows_dev=rd)
    ref = torch.nn.functional.cross_entropy(logits[:R, :V].float().cpu(), labels[:R], ignore_index=-100, reduction="sum")
    assert abs(acc[0].item() - ref.item()) < tol(dt) * abs(ref.item()) and acc[1].item() == R - 1
    keep = logits.clone()
    d = ops.ce_bwd(logits, V, labels.cuda(), lse, acc, rows_dev=rd)
    lg = keep[:R, :V].float().cpu().requires_grad_(True)
    torch.nn.functional.cross_entropy(lg, labels[:R], ignore_index=-100).backward()
    assert rel(d[:R, :V].float().cpu(), lg.grad) < tol(dt) * 2
    assert bool(torch.isnan(d[R:].float()).all())                       # untouched


def test_skinny_accum_and_layernorm_from_accumulator(ops):
    """Decode tails: the reduction of A W^T split over workgroups, every k-slice into a slab of its own (mvlt_gemm_skinny_accum:
    no atomics), then LayerNorm(sum of the slabs + bias + residual) (mvlt_layernorm_acc_fwd) -- bit-reproducible."""
    for dt in DT:
        M, N, K = 64, 768, 3072
        a = rnd((M, K), dt, 12, K ** -0.5); w = rnd((N, K), dt, 13)
        bias = torch.randn(N).cuda(); res = rnd((M, N), dt, 14)
        g = (1.0 + 0.1 * torch.randn(N)).cuda(); b = (0.1 * torch.randn(N)).cuda()
        for splits in (1, 2, 4):
            acc = torch.full((splits, M, N), float("nan"), device="cuda")
            ops.gemm_skinny_accum(a, w, acc, splits)
            ref = a.float() @ w.float().t()
            assert rel(acc.sum(0), ref) < tol(dt)
            y = ops.layernorm_acc_fwd(acc, bias, res, g, b, 1e-12, dt)
            yr = torch.nn.functional.layer_norm(ref + bias + res.float(), (N,), g, b, 1e-12)
            assert rel(y, yr) < tol(dt) * 2
            acc2 = torch.empty_like(acc)
            ops.gemm_skinny_accum(a, w, acc2, splits)
            assert torch.equal(acc, acc2) and torch.equal(y, ops.layernorm_acc_fwd(acc2, bias, res, g, b, 1e-12, dt))


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("res,C_,shift,B", [(14, 96, 3, 2), (28, 192, 0, 1), (14, 192, 3, 3), (14, 384, 3, 1), (14, 128, 0, 2)])
def test_swin_wmsa_fused_backward(ops, dt, res, C_, shift, B):
    """mvlt_swin_wmsa_bwd (proj dgrad + window-attention backward + qkv dgrad in one launch) against the three-launch
    kernel sequence it replaces, on what the fused forward saves."""
    from mvlt_amd._lib import ATTN_SWIN
    from mvlt_amd.indexing import batched_window_maps
    nH = C_ // 32
    if not ops.swin_wmsa_bwd_supported(dt, C_, nH):
        pytest.skip("width not covered by the fused backward kernel in this dtype (LDS)")
    nW = (res // 7) ** 2
    rows = B * res * res
    x = rnd((rows, C_), dt, 70)
    g1 = (1.0 + 0.1 * torch.randn(C_, generator=torch.Generator().manual_seed(71))).cuda()
    b1 = (0.1 * torch.randn(C_, generator=torch.Generator().manual_seed(72))).cuda()
    wqkv = rnd((3 * C_, C_), dt, 73, C_ ** -0.5)
    bqkv = (0.1 * torch.randn(3 * C_, generator=torch.Generator().manual_seed(74))).cuda()
    wproj = rnd((C_, C_), dt, 75, C_ ** -0.5)
    bproj = (0.1 * torch.randn(C_, generator=torch.Generator().manual_seed(76))).cuda()
    table = (0.5 * torch.randn(169, nH, generator=torch.Generator().manual_seed(77))).cuda()
    scale = 32 ** -0.5
    w2n, n2w = batched_window_maps(B, res, res, 7, shift, x.device)
    y, (xn, qkv, ao, lse, mean, rstd) = ops.swin_wmsa_fwd(x, w2n, B, res, nH, shift, g1, b1, 1e-5, wqkv, bqkv, wproj, bproj,
                                                      table, scale, save=True)
    dyw = rnd((rows, C_), dt, 78)
    # the unfused sequence
    dao_u = ops.gemm(dyw, wproj, b_kmajor=True)
    dt_u = torch.zeros(169, nH, device="cuda")
    dqkv_u = ops.attn_bwd(dao_u, qkv, ao, lse, ATTN_SWIN, B * nW, 49, nH, 32, scale, dbias_table=dt_u, bias_table=table,
                          nW=nW, win_res=res, shift=shift)
    dxn_u = ops.gemm(dqkv_u, wqkv, b_kmajor=True)
    # fused
    dt_f = torch.zeros(169, nH, device="cuda")
    dqkv_f, dxn_f = ops.swin_wmsa_bwd(dyw, qkv, lse, B, res, nH, shift, wproj.t().contiguous(), wqkv.t().contiguous(),
                                      table, scale, dt_f)
    t = tol(dt) * 2
    assert rel(dqkv_f, dqkv_u) < t, rel(dqkv_f, dqkv_u)
    assert rel(dxn_f, dxn_u) < t, rel(dxn_f, dxn_u)
    assert rel(dt_f, dt_u) < t, rel(dt_f, dt_u)
