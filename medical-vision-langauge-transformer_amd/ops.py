"""Tensor-level wrappers over the C-ABI (include/mvlt_hip.h).

PyTorch is plumbing here: device memory (caching allocator), the current HIP
stream and dtype bookkeeping.  Every function launches hand-written gfx950
kernels on torch's current stream; nothing falls back to eager torch math.
"""
import ctypes as C
import os

import torch

from . import _lib as L

_DT = {torch.float32: L.F32, torch.bfloat16: L.BF16}
_ws = {}
_ws_need = {}     # (dtype, M, N, K, a_kmajor, b_kmajor, split_k) -> split-K workspace bytes
GEMM_TIMER = None   # bench.py: callable(flops, key) -> (start_event, end_event) or None; .layout = (a_kmajor, b_kmajor) it watches


_stream_cache = [None, None]

# ----------------------------------------------------------------------------- native host path (csrc/host.cpp)
# One native call per SwinTransformerBlock / BertLayer forward or backward pass (torch C++ extension over the same
# C-ABI).  MVLT_NATIVE_HOST=0 keeps every launch on the Python + ctypes path below (same kernels, same results).
NATIVE = os.environ.get("MVLT_NATIVE_HOST", "1") != "0"
_host = None


def host():
    global _host
    if _host is None:
        import importlib.util
        L.lib()                                   # same libmvlt_hip.so the extension binds by rpath
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_mvlt_host.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with `make -C medical-vision-langauge-transformer_amd/csrc` "
                               "(python __graft_entry__.py); there is no fallback")
        spec = importlib.util.spec_from_file_location("_mvlt_host", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        # abi_version() / struct_sizes() are compile-time constants of the extension: a stale _mvlt_host.so (built
        # against an older header) fails here instead of handing truncated structs to the kernels
        if mod.abi_version() != L.ABI_VERSION or list(mod.struct_sizes()) != [C.sizeof(s) for s in L.STRUCTS]:
            raise RuntimeError(f"_mvlt_host.so was compiled against ABI {mod.abi_version()}, libmvlt_hip.so / _lib.py are "
                               f"at ABI {L.ABI_VERSION} (or a struct size differs): rebuild both (make -C .../csrc)")
        _host = mod
    return _host


def stream_int():
    """Handle of the stream the current pass launches on, as a plain int (0 = the null stream)."""
    return int(_stream().value or 0)


def side_int(device):
    return int(side_stream(device).cuda_stream)


def s64(v):
    """64-bit unsigned seed -> the int64 the extension takes (two's complement, cast back in C++)."""
    return v - (1 << 64) if v >= (1 << 63) else v


def _stream():
    # torch.cuda.current_stream() costs ~5 us; engines pin it for the duration of a pass (pin_stream)
    if _stream_cache[0] is not None:
        return _stream_cache[0]
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_side = {}


def side_stream(device):
    """One extra HIP stream per device for work that is off the critical path (weight gradients)."""
    st = _side.get(device.index)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _side[device.index] = st
    return st


class on_side:
    """Launch the enclosed kernels on the side stream once everything queued on the main stream so
    far is done; tensors read there must be passed so the caching allocator keeps them alive."""

    def __init__(self, device, *tensors):
        self.dev, self.tensors = device, tensors

    def __enter__(self):
        side = side_stream(self.dev)
        side.wait_stream(torch.cuda.current_stream())
        # keep the operands alive until the side stream has been joined (cheaper than record_stream per
        # tensor, and also correct under hipGraph capture where the allocator cannot see stream use)
        _side_keepalive.append(self.tensors)
        self.prev = (_stream_cache[0], _stream_cache[1])
        _stream_cache[0], _stream_cache[1] = C.c_void_p(side.cuda_stream), "side"

    def __exit__(self, *a):
        _stream_cache[0], _stream_cache[1] = self.prev


_side_keepalive = []
_opt = {}


def opt_stream(device):
    """Third HIP stream: AdamW of finished gradient buckets runs here while the backward pass continues."""
    st = _opt.get(device.index)
    if st is None:
        st = _opt[device.index] = torch.cuda.Stream(device=device)
    return st


class on_stream:
    """Context manager: launch the enclosed ops on ``stream`` (ordering is the caller's business)."""

    def __init__(self, stream, tag):
        self.stream, self.tag = stream, tag

    def __enter__(self):
        self.prev = (_stream_cache[0], _stream_cache[1])
        _stream_cache[0], _stream_cache[1] = C.c_void_p(self.stream.cuda_stream), self.tag

    def __exit__(self, *a):
        _stream_cache[0], _stream_cache[1] = self.prev


def join_side(device):
    """Main stream waits for the side stream (end of a backward pass / before communication).
    The operands kept alive for the side stream are released to the main-stream allocator only now:
    anything that reuses their memory is ordered after this join."""
    st = _side.get(device.index)
    if st is not None:
        torch.cuda.current_stream().wait_stream(st)
    _side_keepalive.clear()
    if _host is not None:
        _host.side_release()


_pinned_scratch = []


def pin_scratch():
    """A HIP graph is about to record (or has recorded) the addresses of the scratch workspaces: never free them.
    Idempotent (a buffer is pinned once, by address) and limited to the buffers a capture can have seen: the "graph"-tagged
    Python workspaces and the native host path's scratch (the training pass's split-K workspaces used to be pinned
    again at every capture)."""
    have = {b.data_ptr() for b in _pinned_scratch}
    for key, buf in _ws.items():
        if key[2] == "graph" and buf.data_ptr() not in have:
            _pinned_scratch.append(buf)
            have.add(buf.data_ptr())
    if _host is not None:
        _host.scratch_pin()


class pin_stream:
    """Context manager: resolve torch's current stream once for a whole forward/backward pass."""

    def __enter__(self):
        self.prev = (_stream_cache[0], _stream_cache[1])
        # (inside a HIP-graph capture the scratch buffers keep their "graph" tag: their addresses are recorded)
        _stream_cache[0], _stream_cache[1] = C.c_void_p(torch.cuda.current_stream().cuda_stream), ("graph" if self.prev[1] == "graph" else None)

    def __exit__(self, *a):
        _stream_cache[0], _stream_cache[1] = self.prev


_ws_graph_keep = []


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype {t.dtype} (float32 / bfloat16 only)")


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("mvlt_amd ops run on the GPU only (no CPU fallback)")


def workspace(name, nbytes, device):
    key = (name, device.index, _stream_cache[1])      # the side stream gets its own scratch buffers
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None and _stream_cache[1] == "graph":
            _ws_graph_keep.append(buf)                 # a captured HIP graph has this pointer baked in: never free it
        if buf is not None and _stream_cache[1] == "side":
            buf.record_stream(side_stream(device))     # a queued side-stream kernel may still be using it
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        if _stream_cache[1] == "side":
            buf.record_stream(side_stream(device))
        _ws[key] = buf
    return buf


# ----------------------------------------------------------------------------- GEMM
def _ld(t):
    """Row stride of a 2-D operand; torch leaves the stride of a size-1 dimension arbitrary (a one-row matrix
    sliced out of a batch, B = 1 decode), so one-row matrices report their width."""
    return t.stride(0) if t.shape[0] != 1 else max(t.shape[1], 1)


def gemm(A, B, *, a_kmajor=False, b_kmajor=False, out=None, out_f32=False, bias=None, gelu=False,
         save_pre=None, dropout=None, rowscale=None, residual=None, rowmap=None, mul_gelu_grad=None,
         accumulate=False, split_k=0, ldc=None, a_colsum=None, m_dev=None, prefetch=None):
    """C = epilogue(A @ B); see MvltGemm.  A: [M,K] (or [K,M] if a_kmajor);
    B: [N,K] torch-Linear layout (or [K,N] if b_kmajor).
    (Written for a short host path: ~330 calls per training step; pointers go into the struct as plain ints.)"""
    if not (A.is_cuda and B.is_cuda):
        raise RuntimeError("mvlt_amd ops run on the GPU only (no CPU fallback)")
    dtA = A.dtype
    sa, sb = A.shape, B.shape
    assert len(sa) == 2 and len(sb) == 2 and dtA == B.dtype and A.stride(1) == 1 and B.stride(1) == 1
    if a_kmajor:
        K, M = sa
    else:
        M, K = sa
    if b_kmajor:
        Kb, N = sb
    else:
        N, Kb = sb
    assert K == Kb, f"inner dims differ: {K} vs {Kb}"
    odt = torch.float32 if out_f32 else dtA
    if out is None:
        rows = M if rowmap is None else int(rowmap.numel())
        out = torch.empty((rows, N if ldc is None else ldc), dtype=odt, device=A.device)
    assert out.dtype == odt and out.stride(-1) == 1
    p = L.MvltGemm()
    p.dtype = dti = _DT[dtA]
    p.M, p.N, p.K = M, N, K
    p.A, p.lda = A.data_ptr(), (A.stride(0) if sa[0] != 1 else sa[1])
    p.B, p.ldb = B.data_ptr(), (B.stride(0) if sb[0] != 1 else sb[1])
    if a_kmajor:
        p.a_kmajor = 1
    if b_kmajor:
        p.b_kmajor = 1
    p.C = out.data_ptr()
    p.ldc = pldc = (_ld(out) if out.dim() == 2 else N)
    epi = 0
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N
        epi |= L.EPI_BIAS
        p.bias = bias.data_ptr()
    if gelu:
        epi |= L.EPI_GELU
        if save_pre is not None:
            assert save_pre.dtype == dtA and _ld(save_pre) == pldc
            epi |= L.EPI_SAVE_PRE
            p.pre = save_pre.data_ptr()
    if dropout is not None and dropout[0] > 0.0:
        epi |= L.EPI_DROPOUT
        p.dropout_p, p.seed, p.tag = float(dropout[0]), int(dropout[1]), int(dropout[2])
    if rowscale is not None:
        epi |= L.EPI_ROWSCALE
        p.rowscale, p.rows_per_scale = rowscale[0].data_ptr(), int(rowscale[1])
    if residual is not None:
        assert residual.dtype == dtA and residual.stride(-1) == 1
        epi |= L.EPI_RESIDUAL
        p.residual, p.ldr = residual.data_ptr(), _ld(residual)
    if rowmap is not None:
        assert rowmap.dtype == torch.int32
        epi |= L.EPI_ROWMAP
        p.rowmap = rowmap.data_ptr()
    if mul_gelu_grad is not None:
        assert mul_gelu_grad.dtype == dtA and _ld(mul_gelu_grad) == pldc
        epi |= L.EPI_MUL_GELU_GRAD
        p.aux = mul_gelu_grad.data_ptr()
    if out_f32:
        epi |= L.EPI_OUT_F32
    if accumulate:
        epi |= L.EPI_ACCUM
    p.epilogue = epi
    p.split_k = split_k
    if m_dev is not None:          # valid storage rows of A on the device (ragged batch planned on the GPU)
        p.m_dev = m_dev.data_ptr()
    if prefetch is not None:             # weights of the product that runs next: pulled towards the caches by this launch
        p.prefetch, p.prefetch_bytes = prefetch.data_ptr(), prefetch.numel() * prefetch.element_size()
    if a_colsum is not None:
        assert a_kmajor and a_colsum.dtype == torch.float32 and a_colsum.numel() == M
        p.a_colsum = a_colsum.data_ptr()
    lib = L.lib()
    wkey = (dti, M, N, K, a_kmajor, b_kmajor, split_k)
    need = _ws_need.get(wkey)
    if need is None:
        need = _ws_need[wkey] = lib.mvlt_gemm_workspace_bytes(C.byref(p))       # pure function of the key
    if need:
        ws = workspace("gemm", need, A.device)
        p.workspace, p.workspace_bytes = ws.data_ptr(), ws.numel()
    if GEMM_TIMER is not None and GEMM_TIMER.layout == (int(a_kmajor), int(b_kmajor)):
        # bench.py: bracket the launch with HIP events on the stream it is launched on
        bm, bn, sp = C.c_int(), C.c_int(), C.c_int()
        lib.mvlt_gemm_plan(C.byref(p), C.byref(bm), C.byref(bn), C.byref(sp))
        evs = GEMM_TIMER(2.0 * M * N * K, (p.dtype, bm.value, bn.value, int(a_kmajor), int(b_kmajor)))
        if evs is not None:
            st = side_stream(A.device) if _stream_cache[1] == "side" else torch.cuda.current_stream()
            if not getattr(evs[1], "_mvlt_created", False):
                evs[1].record(st)                       # creates the underlying hipEvent_t; re-recorded by the library
                evs[1]._mvlt_created = True
            p.event_after_main = C.c_void_p(evs[1].cuda_event)
            evs[0].record(st)
            L.check(lib.mvlt_gemm(C.byref(p), _stream()), "mvlt_gemm")
            return out
    L.check(lib.mvlt_gemm(C.byref(p), _stream()), "mvlt_gemm")
    return out


def gemm_argmax(A, W, bias=None):
    """argmax_n (A @ W^T + bias) and its value, without materialising the logits (greedy decoding).
    A: [M <= 64, K], W: [N, K] -> (int64 [M], f32 [M])."""
    _need_cuda(A, W)
    M, K = A.shape
    N = W.shape[0]
    p = L.MvltGemm()
    p.dtype, p.M, p.N, p.K = _dt(A), M, N, K
    p.A, p.lda, p.B, p.ldb = _p(A), _ld(A), _p(W), _ld(W)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N
        p.epilogue, p.bias = L.EPI_BIAS, _p(bias)
    nblk = (N + 15) // 16
    pv = torch.empty((M, nblk), dtype=torch.float32, device=A.device)
    pi = torch.empty((M, nblk), dtype=torch.int32, device=A.device)
    idx = torch.empty(M, dtype=torch.int64, device=A.device)
    val = torch.empty(M, dtype=torch.float32, device=A.device)
    L.check(L.lib().mvlt_gemm_argmax(C.byref(p), _p(pv), _p(pi), _p(idx), _p(val), _stream()), "mvlt_gemm_argmax")
    return idx, val


def gemm_argmax_greedy(A, W, bias, state):
    """Decoder GEMM + greedy pick + the per-token bookkeeping of greedy_search in two launches (mvlt_gemm_argmax_greedy).
    A: [M <= 64, K] (rows may be strided), W: [N, K]; ``state``: a prepared ``L.MvltGreedyState`` (decode._GreedyGraph)."""
    M, K = A.shape
    N = W.shape[0]
    p = L.MvltGemm()
    p.dtype, p.M, p.N, p.K = _dt(A), M, N, K
    p.A, p.lda, p.B, p.ldb = _p(A), A.stride(0), _p(W), _ld(W)
    if bias is not None:
        p.epilogue, p.bias = L.EPI_BIAS, _p(bias)
    nblk = (N + 15) // 16
    key = ("argmax_parts", M, nblk, A.device.index)
    buf = _argmax_parts.get(key)
    if buf is None:
        buf = _argmax_parts[key] = (torch.empty((M, nblk), dtype=torch.float32, device=A.device),
                                    torch.empty((M, nblk), dtype=torch.int32, device=A.device))
    L.check(L.lib().mvlt_gemm_argmax_greedy(C.byref(p), _p(buf[0]), _p(buf[1]), C.byref(state), _stream()), "mvlt_gemm_argmax_greedy")


_argmax_parts = {}


def gemm_skinny_accum(A, W, acc, k_splits):
    """acc[s] (f32 [k_splits, M, N]) = A[:, k-slice s] @ W[:, k-slice s]^T: the reduction split over k_splits workgroups per
    column tile, every slice in a slab of its own (no atomics; layernorm_acc_fwd adds them in slice order)."""
    _need_cuda(A, W)
    M, K = A.shape
    N = W.shape[0]
    assert acc.dtype == torch.float32 and acc.shape == (k_splits, M, N) and acc.is_contiguous() and A.stride(1) == 1 and W.stride(1) == 1
    p = L.MvltGemm()
    p.dtype, p.M, p.N, p.K = _dt(A), M, N, K
    p.A, p.lda, p.B, p.ldb = A.data_ptr(), _ld(A), W.data_ptr(), _ld(W)
    L.check(L.lib().mvlt_gemm_skinny_accum(C.byref(p), _p(acc), int(k_splits), _stream()), "mvlt_gemm_skinny_accum")
    return acc


def layernorm_acc_fwd(acc, bias, residual, gamma, beta, eps, dtype, out=None):
    """LayerNorm(sum_s acc[s] + bias + residual); acc: the f32 slabs [nsplit, rows, C] of gemm_skinny_accum."""
    nsplit, rows, Cn = acc.shape
    y = out if out is not None else torch.empty((rows, Cn), dtype=dtype, device=acc.device)
    L.check(L.lib().mvlt_layernorm_acc_fwd(_DT[dtype], _p(acc), int(nsplit), _p(bias), _p(residual), _p(gamma), _p(beta), float(eps),
                                           rows, Cn, _p(y), _stream()), "mvlt_layernorm_acc_fwd")
    return y


_GROUP_LONG_K = 8192
# MVLT_DETERMINISTIC=1: no float-atomic k-slices in the GEMM path (the 4-wave fallback of the Swin stage-0/1 weight
# gradients takes split-K slabs + the deterministic reduce).  Greedy decoding needs no flag since round 5: its split
# reductions meet in slabs summed in slice order.  Still
# accumulated with float atomics under the flag: the relative-position-bias-table gradient (swin_attn_bwd2_kernel) and the
# word-embedding gradient (embed_bwd); tests/test_model_gpu.py::test_config2_step_is_bit_reproducible_* lists them.  The
# flag is read once at import: set it before importing the package.
DETERMINISTIC = os.environ.get("MVLT_DETERMINISTIC", "0") == "1"


def wgrad_group(items):
    """Weight gradients of one layer: items = [(dY [R,No], X [R,Ni], dW f32 [No,Ni], dbias f32 [No] | None), ...]
    -> dW_i = dY_i^T X_i (and dbias_i = column sums of dY_i).  When the items qualify (mvlt_gemm_group) and
    their 64-row tiles together fill the GPU, they go out as ONE launch without split-K -- instead of
    len(items) split-K launches plus their reduce kernels; with few tiles but >= 8192 reduction rows (Swin stages
    0/1) still one launch, cut into k-slices that meet in the zeroed f32 output by atomicAdd; otherwise one gemm() each."""
    lib = L.lib()
    n = len(items)
    m_dev = None
    if items and len(items[0]) == 5:          # (dy, x, dw, db, m_dev): the reduction length lives on the device
        m_dev = items[0][4]
        items = [it[:4] for it in items]
    bn = 128 if all(x.shape[1] % 128 == 0 for _, x, _, _ in items) else (96 if all(x.shape[1] % 96 == 0 for _, x, _, _ in items) else 0)
    tiles = sum(((dy.shape[1] + 63) // 64) * ((x.shape[1] + bn - 1) // bn) for dy, x, _, _ in items) if bn else 0
    # few tiles but a long reduction (Swin stages 0/1): still one launch, cut into k-slices inside mvlt_gemm_group
    slices_ok = items[0][0].dtype == torch.bfloat16      # bf16: k-slices through the 8-wave engine's deterministic slab reduce
    if not (1 < n <= 8 and bn and (tiles >= 200 or (slices_ok and items[0][0].shape[0] >= _GROUP_LONG_K))):
        for dy, x, dw, db in items:
            gemm(dy, x, a_kmajor=True, b_kmajor=True, out=dw, out_f32=True, a_colsum=db, m_dev=m_dev)
        return
    arr = (L.MvltGemm * n)()
    flops = 0.0
    for i, (dy, x, dw, db) in enumerate(items):
        _need_cuda(dy, x)
        assert dy.dtype == x.dtype and dy.stride(1) == 1 and x.stride(1) == 1 and dy.shape[0] == x.shape[0]
        assert dw.dtype == torch.float32 and dw.shape == (dy.shape[1], x.shape[1]) and dw.stride(1) == 1
        p = arr[i]
        p.dtype, p.M, p.N, p.K = _dt(dy), dy.shape[1], x.shape[1], dy.shape[0]
        p.A, p.lda, p.a_kmajor = _p(dy), dy.stride(0), 1
        p.B, p.ldb, p.b_kmajor = _p(x), x.stride(0), 1
        p.C, p.ldc = _p(dw), dw.stride(0)
        p.epilogue, p.split_k = L.EPI_OUT_F32, 1
        if m_dev is not None:
            p.m_dev = m_dev.data_ptr()
        if db is not None:
            assert db.dtype == torch.float32 and db.numel() == dy.shape[1]
            p.a_colsum = _p(db)
        flops += 2.0 * p.M * p.N * p.K
    need = lib.mvlt_gemm_group_workspace_bytes(arr, n)          # k-slice slabs of the 8-wave engine (0: none)
    if need:
        ws = workspace("gemm_group", need, items[0][0].device)
        arr[0].workspace, arr[0].workspace_bytes = ws.data_ptr(), ws.numel()
    evs = GEMM_TIMER(flops, ("group", arr[0].dtype, 64, bn)) if GEMM_TIMER is not None else None
    st = None
    if evs is not None:
        st = side_stream(items[0][0].device) if _stream_cache[1] == "side" else torch.cuda.current_stream()
        evs[0].record(st)
    L.check(lib.mvlt_gemm_group(arr, n, _stream()), "mvlt_gemm_group")
    if evs is not None:
        evs[1].record(st)


def prefetch(tensors):
    """Pull up to 8 read-only tensors (weights about to be used) towards the GPU's caches: one launch, nothing written."""
    arr = (L.MvltRange * len(tensors))()
    for i, t in enumerate(tensors):
        assert t.is_cuda and t.is_contiguous()
        arr[i].ptr, arr[i].bytes = t.data_ptr(), t.numel() * t.element_size()
    L.check(L.lib().mvlt_prefetch(arr, len(tensors), _stream()), "mvlt_prefetch")


def zero_batch(tensors):
    """Zero several small f32 tensors in one launch."""
    arr = (L.MvltZeroItem * len(tensors))()
    for i, t in enumerate(tensors):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda
        arr[i].ptr, arr[i].n = t.data_ptr(), t.numel()
    L.check(L.lib().mvlt_zero_batch(arr, len(tensors), _stream()), "mvlt_zero_batch")


def colsum(x, out=None, accumulate=False):
    """out[n] = sum_m x[m,n] (f32) -- bias gradients."""
    _need_cuda(x)
    M, N = x.shape
    lib = L.lib()
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=x.device)
    ws = workspace("colsum", lib.mvlt_colsum_workspace_rows(M) * N * 4, x.device)
    L.check(lib.mvlt_colsum(_dt(x), _p(x), x.stride(0), M, N, _p(out), int(accumulate), _p(ws), _stream()),
            "mvlt_colsum")
    return out


# ----------------------------------------------------------------------------- LayerNorm
def layernorm_fwd(x, gamma, beta, eps, *, rows=None, C_=None, out=None, out_rowmap=None, merge=None, gelu=False,
                  save_pre=False, save_stats=True, rows_dev=None):
    """x: [..., C] contiguous (or the un-merged [B,H*W,C/4] tensor when merge=(H,W))."""
    _need_cuda(x)
    assert x.is_contiguous() and gamma.dtype == torch.float32
    Cn = gamma.numel()
    if merge is None:
        nrows = x.numel() // Cn
        oshape = x.shape
    else:
        H, W = merge
        nrows = x.numel() // Cn
        oshape = (x.shape[0], (H // 2) * (W // 2), Cn)
    y = out if out is not None else torch.empty(oshape, dtype=x.dtype, device=x.device)
    p = L.MvltLayerNorm()
    p.dtype, p.rows, p.C, p.eps = _DT[x.dtype], nrows, Cn, eps
    p.x, p.gamma, p.beta, p.y = x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr()
    mean = rstd = ypre = None
    if save_stats:
        mean, rstd = torch.empty((2, nrows), dtype=torch.float32, device=x.device).unbind(0)
        p.mean = pm = mean.data_ptr()
        p.rstd = pm + 4 * nrows
    if gelu:
        p.gelu = 1
        if save_pre:
            ypre = torch.empty_like(y)
            p.y_pre = _p(ypre)
    if out_rowmap is not None:
        assert out_rowmap.dtype == torch.int32 and out_rowmap.numel() == nrows
        p.out_rowmap = _p(out_rowmap)
    if merge is not None:
        p.merge_H, p.merge_W = merge
    if rows_dev is not None:
        p.rows_dev = rows_dev.data_ptr()
    L.check(L.lib().mvlt_layernorm_fwd(C.byref(p), _stream()), "mvlt_layernorm_fwd")
    return y, mean, rstd, ypre


def layernorm_bwd(dy, x, mean, rstd, gamma, dgamma, dbeta, *, dy_rowmap=None, y_pre=None, dres=None, merge=None,
                  accumulate=False, dx=None, branch=None, defer=None, rows_dev=None, dy_parts=False):
    """branch=dict(rowmap=, rowscale=(t, rps), dropout=(p, seed, tag)) also returns the branch gradient dz.
    defer=LnReduceQueue: the dgamma/dbeta partial rows are reduced later in one batched launch.
    dy_parts=True: dy is [parts, rows, C], 1 .. 4 partial tensors whose sum is the gradient (swin_wmsa2_bwd's dxn_parts)."""
    _need_cuda(dy, x)
    Cn = gamma.numel()
    nrows = mean.numel()
    if dx is None:
        dx = torch.empty_like(x)
    lib = L.lib()
    if defer is not None:
        ws = defer.take(2 * lib.mvlt_layernorm_bwd_workspace_rows() * Cn, x.device)
        defer.items.append((ws, lib.mvlt_layernorm_bwd_nparts(nrows, Cn), Cn, dgamma, dbeta))
        if defer not in LnReduceQueue._active:
            LnReduceQueue._active.append(defer)
    else:
        ws = workspace("ln_bwd", 2 * lib.mvlt_layernorm_bwd_workspace_rows() * Cn * 4, x.device)
    p = L.MvltLayerNormBwd()
    p.dtype, p.rows, p.C = _dt(x), nrows, Cn
    p.dy, p.x, p.mean, p.rstd, p.gamma = _p(dy), _p(x), _p(mean), _p(rstd), _p(gamma)
    if dy_parts:
        assert dy.dim() == 3 and dy.is_contiguous() and dy.shape[1:] == (nrows, Cn)
        if dy.shape[0] > 1:
            p.dy_parts, p.dy_part_stride = dy.shape[0], dy.stride(0)
    if dy_rowmap is not None:
        p.dy_rowmap = _p(dy_rowmap)
    if y_pre is not None:
        p.y_pre, p.gelu = _p(y_pre), 1
    if dres is not None:
        p.dres = _p(dres)
    p.dx = _p(dx)
    if merge is not None:
        p.merge_H, p.merge_W = merge
    p.dgamma, p.dbeta, p.accumulate, p.workspace = _p(dgamma), _p(dbeta), int(accumulate), _p(ws)
    p.defer_param_reduce = int(defer is not None)
    if rows_dev is not None:
        p.rows_dev = rows_dev.data_ptr()
    dz = None
    if branch is not None:
        dz = torch.empty((nrows, Cn), dtype=x.dtype, device=x.device)
        p.dz = _p(dz)
        if branch.get("rowmap") is not None:
            p.dz_rowmap = _p(branch["rowmap"])
        if branch.get("rowscale") is not None:
            p.dz_rowscale, p.dz_rows_per_scale = _p(branch["rowscale"][0]), int(branch["rowscale"][1])
        if branch.get("dropout") is not None and branch["dropout"][0] > 0:
            p.dz_dropout_p, p.seed, p.tag = float(branch["dropout"][0]), int(branch["dropout"][1]), int(branch["dropout"][2])
    L.check(lib.mvlt_layernorm_bwd(C.byref(p), _stream()), "mvlt_layernorm_bwd")
    return dx if branch is None else (dx, dz)


class LnReduceQueue:
    """Partial dgamma/dbeta rows of every LayerNorm of one backward pass live in one pool and are
    reduced by ceil(n/96) launches at the end (instead of one launch per LayerNorm)."""
    _pool = {}
    _active = []          # queues with pending items (flushed early by the DDP bucket launcher)

    def __init__(self):
        self.items, self.off = [], 0

    @staticmethod
    def flush_all():
        """Write every pending dgamma/dbeta now: a gradient bucket is about to be communicated."""
        for q in list(LnReduceQueue._active):
            q.flush()
        if _host is not None and _host.lnq_pending():
            _host.lnq_flush(stream_int())

    def take(self, nfloats, device):
        pool = LnReduceQueue._pool.get(device.index)
        need = self.off + nfloats
        if pool is None or pool.numel() < need:
            newp = torch.empty(max(need, 48 << 20), dtype=torch.float32, device=device)
            if pool is not None:            # keep earlier slices valid: they are still referenced by self.items
                self._keep = getattr(self, "_keep", []) + [pool]
                self.off = 0
            LnReduceQueue._pool[device.index] = pool = newp
        ws = pool[self.off:self.off + nfloats]
        self.off += nfloats
        return ws

    def flush(self):
        if self in LnReduceQueue._active:
            LnReduceQueue._active.remove(self)
        if not self.items:
            return
        arr = (L.MvltLnReduceItem * len(self.items))()
        for i, (ws, nparts, Cn, dg, db) in enumerate(self.items):
            arr[i].workspace, arr[i].nparts, arr[i].C, arr[i].dgamma, arr[i].dbeta = ws.data_ptr(), nparts, Cn, dg.data_ptr(), db.data_ptr()
        L.check(L.lib().mvlt_layernorm_param_reduce_batch(arr, len(self.items), _stream()), "mvlt_layernorm_param_reduce_batch")
        self.items, self.off = [], 0


# ----------------------------------------------------------------------------- attention
def _attn_struct(qkv, out, lse, mode, nseq, Lq, nH, hd, scale, *, bias_table=None, nW=0, win_res=0, shift=0,
                 text_ids=None, image_mask=None, obj_end=0, dropout=None, pack=None):
    p = L.MvltAttn()
    if pack is not None:          # (row_start int32 [nseq], seq_len int32 [nseq], total rows): packed activations
        assert pack[0].dtype == torch.int32 and pack[1].dtype == torch.int32 and pack[0].numel() == nseq
        p.row_start, p.seq_len = _p(pack[0]), _p(pack[1])
    p.dtype, p.mode, p.nseq, p.L, p.nH, p.hd = _dt(qkv), mode, nseq, Lq, nH, hd
    p.qkv, p.out, p.lse, p.scale = _p(qkv), _p(out), _p(lse), float(scale)
    if bias_table is not None:
        assert bias_table.dtype == torch.float32
        p.bias_table, p.nW, p.win_res, p.shift = _p(bias_table), nW, win_res, shift
    if text_ids is not None:
        assert text_ids.dtype == torch.int64 and text_ids.is_contiguous()
        p.text_ids, p.T = _p(text_ids), text_ids.shape[1]
    if image_mask is not None:
        p.image_mask = _p(image_mask)
    p.obj_end = obj_end
    if dropout is not None and dropout[0] > 0.0:
        p.dropout_p, p.seed, p.tag = float(dropout[0]), int(dropout[1]), int(dropout[2])
    return p


def attn_fwd(qkv, mode, nseq, Lq, nH, hd, scale, **kw):
    """qkv: [nseq*L, 3*nH*hd] -> (out [nseq*L, nH*hd], lse [nseq,nH,L]); with pack= the rows are packed
    ([R, ...], sequence s at rows row_start[s] .. +seq_len[s])."""
    _need_cuda(qkv)
    rows = nseq * Lq if kw.get("pack") is None else kw["pack"][2]
    assert qkv.is_contiguous() and qkv.shape == (rows, 3 * nH * hd)
    out = torch.empty((rows, nH * hd), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((nseq, nH, Lq), dtype=torch.float32, device=qkv.device)
    p = _attn_struct(qkv, out, lse, mode, nseq, Lq, nH, hd, scale, **kw)
    L.check(L.lib().mvlt_attn_fwd(C.byref(p), _stream()), "mvlt_attn_fwd")
    return out, lse


def swin_bwd_proj_supported(dtype, nH, hd, shift):
    """MvltAttn.dout_weight (the output projection's dgrad inside the Swin attention backward): bf16, head_dim 32, 3 / 6 / 12 heads."""
    return dtype == torch.bfloat16 and hd == 32 and nH in (3, 6, 12) and shift in (0, 3)


def attn_bwd(dout, qkv, out, lse, mode, nseq, Lq, nH, hd, scale, dbias_table=None, event=None, dout_weight=None, **kw):
    """event: a hipEvent_t handle (int) that completes with the call's last kernel (mvlt_attn_bwd_ev: the fork of the
    weight-gradient stream without a marker packet on this stream).  dout_weight (Swin only): `dout` is the gradient of the output
    projection's OUTPUT and the kernel applies proj.weight's transpose per head itself (MvltAttn.dout_weight)."""
    _need_cuda(qkv, dout)
    assert dout.is_contiguous() and dout.shape == out.shape
    dqkv = torch.empty_like(qkv)
    p = _attn_struct(qkv, out, lse, mode, nseq, Lq, nH, hd, scale, **kw)
    p.dout, p.dqkv = _p(dout), _p(dqkv)
    if dout_weight is not None:
        assert mode == L.ATTN_SWIN and dout_weight.dtype == qkv.dtype and dout_weight.is_contiguous() and dout_weight.shape == (nH * hd, nH * hd)
        p.dout_weight = _p(dout_weight)
    if mode != L.ATTN_SWIN:
        delta = torch.empty_like(lse)          # delta_q hand-over between the two backward launches
        p.delta_ws = _p(delta)
    if dbias_table is not None:
        assert dbias_table.dtype == torch.float32
        p.dbias_table = _p(dbias_table)
    if event is not None:
        L.check(L.lib().mvlt_attn_bwd_ev(C.byref(p), _stream(), C.c_void_p(event)), "mvlt_attn_bwd_ev")
    else:
        L.check(L.lib().mvlt_attn_bwd(C.byref(p), _stream()), "mvlt_attn_bwd")
    return dqkv


# ----------------------------------------------------------------------------- fused Swin W-MSA
def swin_wmsa_supported(dtype, C_, nH):
    return bool(L.lib().mvlt_swin_wmsa_supported(_DT[dtype], C_, nH))


def _wmsa_struct(x, w2n, B, res, nH, shift, gamma, beta, eps, wqkv, bqkv, wproj, bproj, table, scale, rowscale):
    Cn = x.shape[1]
    assert x.is_contiguous() and x.shape[0] == B * res * res and w2n.dtype == torch.int32 and w2n.numel() == x.shape[0]
    assert wqkv.dtype == x.dtype and wproj.dtype == x.dtype and wqkv.shape == (3 * Cn, Cn) and wproj.shape == (Cn, Cn)
    assert wqkv.is_contiguous() and wproj.is_contiguous() and table.dtype == torch.float32 and table.shape == (169, nH)
    p = L.MvltSwinWmsa()
    p.dtype, p.B, p.res, p.C, p.nH, p.shift = _DT[x.dtype], B, res, Cn, nH, shift
    p.x, p.w2n = x.data_ptr(), w2n.data_ptr()
    p.ln_gamma, p.ln_beta, p.ln_eps = gamma.data_ptr(), beta.data_ptr(), eps
    p.wqkv, p.bqkv, p.wproj, p.bproj = wqkv.data_ptr(), bqkv.data_ptr(), wproj.data_ptr(), bproj.data_ptr()
    p.bias_table, p.scale = table.data_ptr(), float(scale)
    if rowscale is not None:
        assert rowscale.dtype == torch.float32 and rowscale.numel() == B and rowscale.is_contiguous()
        p.rowscale = rowscale.data_ptr()
    return p


def swin_wmsa_fwd(x, w2n, B, res, nH, shift, gamma, beta, eps, wqkv, bqkv, wproj, bproj, table, scale,
                  rowscale=None, save=False, head_split=False):
    """y = x + rowscale * proj(window_attention(qkv(norm1(x))))  (MvltSwinWmsa): one launch.
    save=True also returns (xn_win, qkv_win, attn_out, lse, mean, rstd) for the backward pass.
    head_split=True: one workgroup per (window, head group), no output projection: returns (attn_out, saved)."""
    _need_cuda(x)
    p = _wmsa_struct(x, w2n, B, res, nH, shift, gamma, beta, eps, wqkv, bqkv, wproj, bproj, table, scale, rowscale)
    if head_split:
        p.head_split = 1
        ao = torch.empty_like(x)
        p.attn_out = ao.data_ptr()
        saved = None
        if save:
            rows, Cn = x.shape
            xn = torch.empty_like(x)
            qkv = torch.empty((rows, 3 * Cn), dtype=x.dtype, device=x.device)
            lse = torch.empty((rows // 49, nH, 49), dtype=torch.float32, device=x.device)
            mean, rstd = torch.empty((2, rows), dtype=torch.float32, device=x.device).unbind(0)
            p.xn_win, p.lse, p.mean, p.rstd, p.qkv_win = xn.data_ptr(), lse.data_ptr(), mean.data_ptr(), rstd.data_ptr(), qkv.data_ptr()
            saved = (xn, qkv, ao, lse, mean, rstd)
        L.check(L.lib().mvlt_swin_wmsa_fwd(C.byref(p), _stream()), "mvlt_swin_wmsa_fwd")
        return ao, saved
    y = torch.empty_like(x)
    p.y = y.data_ptr()
    saved = None
    if save:
        rows, Cn = x.shape
        xn, ao = torch.empty_like(x), torch.empty_like(x)
        qkv = torch.empty((rows, 3 * Cn), dtype=x.dtype, device=x.device)
        lse = torch.empty((rows // 49, nH, 49), dtype=torch.float32, device=x.device)
        mean, rstd = torch.empty((2, rows), dtype=torch.float32, device=x.device).unbind(0)
        p.xn_win, p.attn_out, p.lse, p.mean, p.rstd = xn.data_ptr(), ao.data_ptr(), lse.data_ptr(), mean.data_ptr(), rstd.data_ptr()
        p.qkv_win = qkv.data_ptr()
        saved = (xn, qkv, ao, lse, mean, rstd)
    L.check(L.lib().mvlt_swin_wmsa_fwd(C.byref(p), _stream()), "mvlt_swin_wmsa_fwd")
    return y, saved


def swin_wmsa2_supported(dtype, B, res, C_, nH):
    return bool(L.lib().mvlt_swin_wmsa2_supported(_DT[dtype], B, res, C_, nH))


_WMSA2_SYNC = {}          # (device, stream) -> int32 hand-off workspace
_WMSA2_PINNED = {}        # key -> (pinned int32 [1], event): asynchronous copies of the error count


def wmsa2_sync_ws(device, words):
    """The int32 hand-off workspace of mvlt_swin_wmsa2_fwd for the CURRENT stream of `device`: zeroed once here, every
    launch leaves its counters zeroed; word 0 is the sticky error count (include/mvlt_hip.h).  One workspace per (device,
    stream): launches of one stream are serialised, launches of different streams never share counters."""
    st = _stream_cache[0]          # (pinned by the engines for the duration of a pass: torch.cuda.current_stream costs ~5 us)
    if _stream_cache[1] == "graph":
        # ONE workspace for every captured inference graph of the device (replays are serialised on the caller's stream): it is
        # allocated and zeroed by the eager warm-up run that precedes a capture (runtime.GraphedEval._capture), so no memset is
        # ever captured -- a replay must not clear the sticky error count -- and it is never reallocated under a graph that has
        # its address baked in (a larger one is a NEW buffer; the old one is kept alive)
        key = (torch.device(device), "graph")
    else:
        key = (torch.device(device), (st.value or 0) if st is not None else torch.cuda.current_stream(device).cuda_stream)
    ws = _WMSA2_SYNC.get(key)
    if ws is None or ws.numel() < words:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("mvlt_amd: the W-MSA hand-off workspace must exist before a HIP-graph capture (run the call once "
                               "eagerly under ops.on_stream(stream, 'graph') first, as runtime.GraphedEval does)")
        if ws is not None and key[1] == "graph":
            _ws_graph_keep.append(ws)
        ws = torch.zeros(max(words, 4096), dtype=torch.int32, device=device)
        _WMSA2_SYNC[key] = ws
    return ws


class DeviceHandoffError(RuntimeError):
    """A bounded in-launch wait of mvlt_swin_wmsa2_fwd ran out: the head groups of a window pair were not co-resident
    (something else held the CUs / the LDS).  The affected rows of the block output are NaN."""


def _wmsa2_raise(n):
    raise DeviceHandoffError(
        f"mvlt_swin_wmsa2_fwd: {n} hand-off wait(s) between head-group workgroups ran out (the groups of a window pair were "
        "not resident together -- another kernel or process was holding CUs / LDS).  The affected rows of the block output "
        "were written as NaN.  Set MVLT_WMSA2=0 to use the kernels without an in-launch hand-off.")


def wmsa2_sync_errors():
    """Sum of the sticky error counts of all hand-off workspaces (0 = no bounded wait ever ran out).  Forces a device sync."""
    return sum(int(ws[0].item()) for ws in _WMSA2_SYNC.values())


def wmsa2_check(sync=True):
    """Raise DeviceHandoffError if a hand-off wait ran out.  sync=True reads the error counts now (a device sync: use it
    where the caller synchronises anyway -- loss.item(), save_pretrained).  sync=False costs no sync: it looks at the copies
    queued by the PREVIOUS call (pinned host memory, ready when their event has completed) and queues fresh ones behind
    the work issued so far, so a failure surfaces one call late."""
    if sync:
        n = wmsa2_sync_errors()
        if n:
            _wmsa2_raise(n)
        return
    for key, ws in _WMSA2_SYNC.items():
        slot = _WMSA2_PINNED.get(key)
        if slot is None:
            slot = _WMSA2_PINNED[key] = [torch.zeros(1, dtype=torch.int32).pin_memory(), None]
        host, ev = slot
        if ev is not None and ev.query() and int(host[0]):
            _wmsa2_raise(int(host[0]))
        if ev is None or ev.query():          # (never rewrite the pinned word while a copy into it is in flight)
            host.copy_(ws[:1], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(key[0]))
            slot[1] = ev


def wmsa2_error_tensor(device):
    """f32 [1] on `device`: the sum of the sticky error counts of its hand-off workspaces (no sync), or None if the fused W-MSA
    kernel has not run there.  ddp.GradReducer sends it along with the label count so that EVERY rank learns of a time-out."""
    dev = torch.device(device)
    ws = [w for k, w in _WMSA2_SYNC.items() if k[0] == dev]
    if not ws:
        return None
    return torch.stack([w[0] for w in ws]).sum().to(torch.float32).reshape(1)


def wmsa2_clear_errors():
    """Zero the error counts and the counters (after a reported failure; the caller has synchronised)."""
    torch.cuda.synchronize()
    for ws in _WMSA2_SYNC.values():
        ws.zero_()
    for slot in _WMSA2_PINNED.values():
        slot[0].zero_()
        slot[1] = None


def wmsa2_set_timeout_ms(ms):
    L.check(L.lib().mvlt_swin_wmsa2_set_timeout_ms(int(ms)), "mvlt_swin_wmsa2_set_timeout_ms")


def debug_stream_copy(dst, src, blocks=32, inflight=2, stream=None):
    """Diagnostic: copy src -> dst (same byte count, may alias) with the launch geometry of a ring collective's kernel."""
    _need_cuda(dst, src)
    nbytes = src.numel() * src.element_size()
    assert dst.numel() * dst.element_size() == nbytes and nbytes % 16 == 0
    st = C.c_void_p(stream.cuda_stream) if stream is not None else _stream()
    L.check(L.lib().mvlt_debug_stream_copy(_p(dst), _p(src), nbytes, int(blocks), int(inflight), st), "mvlt_debug_stream_copy")


def debug_hold_cus(blocks, lds_bytes, usec, stream=None):
    """Diagnostic: occupy CUs with spinning workgroups on `stream` (default: the current one)."""
    st = stream.cuda_stream if stream is not None else _stream()
    L.check(L.lib().mvlt_debug_hold_cus(int(blocks), int(lds_bytes), int(usec), st), "mvlt_debug_hold_cus")


def swin_wmsa2_fwd(x, w2n, B, res, nH, shift, gamma, beta, eps, wqkv, bqkv, wproj, bproj, table, scale,
                   rowscale=None, save=False):
    """Second design of the fused forward (csrc/wmsa2.hip): same result as swin_wmsa_fwd, two windows per workgroup,
    head groups across workgroups.  save=True also returns (xn_win, qkv_win, attn_out, lse, mean, rstd)."""
    _need_cuda(x)
    p = _wmsa_struct(x, w2n, B, res, nH, shift, gamma, beta, eps, wqkv, bqkv, wproj, bproj, table, scale, rowscale)
    y = torch.empty_like(x)
    ao = torch.empty_like(x)
    p.y, p.attn_out = y.data_ptr(), ao.data_ptr()
    saved = None
    if save:
        rows, Cn = x.shape
        xn = torch.empty_like(x)
        qkv = torch.empty((rows, 3 * Cn), dtype=x.dtype, device=x.device)
        lse = torch.empty((rows // 49, nH, 49), dtype=torch.float32, device=x.device)
        mean, rstd = torch.empty((2, rows), dtype=torch.float32, device=x.device).unbind(0)
        p.xn_win, p.lse, p.mean, p.rstd, p.qkv_win = xn.data_ptr(), lse.data_ptr(), mean.data_ptr(), rstd.data_ptr(), qkv.data_ptr()
        saved = (xn, qkv, ao, lse, mean, rstd)
    ws = wmsa2_sync_ws(x.device, L.lib().mvlt_swin_wmsa2_sync_words(B, res))
    L.check(L.lib().mvlt_swin_wmsa2_fwd(C.byref(p), ws.data_ptr(), _stream()), "mvlt_swin_wmsa2_fwd")
    return y, saved


def swin_wmsa_bwd_supported(dtype, C_, nH):
    return bool(L.lib().mvlt_swin_wmsa_bwd_supported(_DT[dtype], C_, nH))


def swin_wmsa_bwd(dy_win, qkv_win, lse, B, res, nH, shift, wproj_t, wqkv_t, table, scale, dtable):
    """Output-projection dgrad + window-attention backward + qkv dgrad in one launch (mvlt_swin_wmsa_bwd).
    dy_win [rows, C] window order (DropPath scale applied); wproj_t = proj.weight^T, wqkv_t = qkv.weight^T (compute
    dtype, contiguous).  Returns (dqkv [rows, 3C], dxn_win [rows, C]); dtable (f32 [169, nH]) is accumulated."""
    _need_cuda(dy_win)
    rows, Cn = dy_win.shape
    assert dy_win.is_contiguous() and qkv_win.is_contiguous() and qkv_win.shape == (rows, 3 * Cn)
    assert wproj_t.is_contiguous() and wproj_t.shape == (Cn, Cn) and wqkv_t.is_contiguous() and wqkv_t.shape == (Cn, 3 * Cn)
    assert lse.dtype == torch.float32 and table.dtype == torch.float32 and dtable.dtype == torch.float32
    p = L.MvltSwinWmsa()
    p.dtype, p.B, p.res, p.C, p.nH, p.shift = _DT[dy_win.dtype], B, res, Cn, nH, shift
    p.dy_win, p.qkv_win, p.lse = dy_win.data_ptr(), qkv_win.data_ptr(), lse.data_ptr()
    p.wproj_t, p.wqkv_t, p.bias_table, p.scale = wproj_t.data_ptr(), wqkv_t.data_ptr(), table.data_ptr(), float(scale)
    dqkv = torch.empty_like(qkv_win)
    dxn = torch.empty_like(dy_win)
    p.dqkv, p.dxn_win, p.dbias_table = dqkv.data_ptr(), dxn.data_ptr(), dtable.data_ptr()
    L.check(L.lib().mvlt_swin_wmsa_bwd(C.byref(p), _stream()), "mvlt_swin_wmsa_bwd")
    return dqkv, dxn


def swin_wmsa2_bwd_parts(dtype, B, res, C_, nH):
    """Number of partial dXn tensors the one-launch backward of the second design writes (nH / 3); 0 = shape not covered."""
    return int(L.lib().mvlt_swin_wmsa2_bwd_parts(_DT[dtype], B, res, C_, nH))


def swin_dbias_reduce(items):
    """items: [(ws f32 [nwg, 3 * 169], nH, dtable f32 [169, nH])]: the per-workgroup sums of mvlt_swin_wmsa2_bwd launches added
    into their tables, one launch for all of them (mvlt_swin_wmsa2_bwd_dbias)."""
    arr = (L.MvltSwinDbiasItem * len(items))()
    for i, (ws, nH, dtable) in enumerate(items):
        assert ws.dtype == torch.float32 and ws.is_contiguous() and ws.shape[1] == 507 and dtable.dtype == torch.float32 and dtable.shape == (169, nH)
        arr[i].ws, arr[i].nwg, arr[i].nH, arr[i].dbias_table = ws.data_ptr(), ws.shape[0], nH, dtable.data_ptr()
    L.check(L.lib().mvlt_swin_wmsa2_bwd_dbias(arr, len(items), _stream()), "mvlt_swin_wmsa2_bwd_dbias")


def swin_wmsa2_bwd(dy_win, qkv_win, lse, B, res, nH, shift, wproj, wqkv, table, scale, dtable, event=None, dbias_defer=None):
    """Output-projection dgrad + window-attention backward + qkv dgrad in one launch, second design (mvlt_swin_wmsa2_bwd).
    dy_win [rows, C] window order (DropPath scale applied); wproj [C, C], wqkv [3C, C] as stored (compute dtype).
    Returns (dqkv [rows, 3C], dxn_parts [nH / 3, rows, C]): the qkv dgrad is the SUM of the parts (layernorm_bwd adds them
    while it loads: dy_parts); dtable (f32 [169, nH]) is accumulated -- by a second small launch here, or later together with
    other blocks' when dbias_defer (a list) is given: it receives (ws, nH, dtable) for swin_dbias_reduce."""
    _need_cuda(dy_win)
    rows, Cn = dy_win.shape
    nparts = swin_wmsa2_bwd_parts(dy_win.dtype, B, res, Cn, nH)
    assert nparts > 0, "mvlt_swin_wmsa2_bwd: shape not covered"
    assert dy_win.is_contiguous() and qkv_win.is_contiguous() and qkv_win.shape == (rows, 3 * Cn) and rows == B * res * res
    assert wproj.is_contiguous() and wproj.shape == (Cn, Cn) and wqkv.is_contiguous() and wqkv.shape == (3 * Cn, Cn)
    assert wproj.dtype == dy_win.dtype and wqkv.dtype == dy_win.dtype
    assert lse.dtype == torch.float32 and table.dtype == torch.float32 and dtable.dtype == torch.float32
    p = L.MvltSwinWmsa()
    p.dtype, p.B, p.res, p.C, p.nH, p.shift = _DT[dy_win.dtype], B, res, Cn, nH, shift
    p.dy_win, p.qkv_win, p.lse = dy_win.data_ptr(), qkv_win.data_ptr(), lse.data_ptr()
    p.wproj, p.wqkv, p.bias_table, p.scale = wproj.data_ptr(), wqkv.data_ptr(), table.data_ptr(), float(scale)
    dqkv = torch.empty_like(qkv_win)
    dxn = torch.empty((nparts, rows, Cn), dtype=dy_win.dtype, device=dy_win.device)
    p.dqkv, p.dxn_win = dqkv.data_ptr(), dxn.data_ptr()
    ws = None
    if dtable is not None:
        nwg = int(L.lib().mvlt_swin_wmsa2_bwd_workgroups(p.dtype, B, res, Cn, nH))
        ws = torch.empty((nwg, 507), dtype=torch.float32, device=dy_win.device)
        p.dbias_ws = ws.data_ptr()
    if event is not None:
        L.check(L.lib().mvlt_swin_wmsa2_bwd_ev(C.byref(p), _stream(), C.c_void_p(event)), "mvlt_swin_wmsa2_bwd_ev")
    else:
        L.check(L.lib().mvlt_swin_wmsa2_bwd(C.byref(p), _stream()), "mvlt_swin_wmsa2_bwd")
    if ws is not None:
        if dbias_defer is not None:
            dbias_defer.append((ws, nH, dtable))
        else:
            swin_dbias_reduce([(ws, nH, dtable)])
    return dqkv, dxn


# ----------------------------------------------------------------------------- data movement
def im2col_patch(img, dtype, patch):
    _need_cuda(img)
    assert img.dtype == torch.float32 and img.is_contiguous() and img.dim() == 4 and img.shape[2] == img.shape[3]
    B, Cin, S, _ = img.shape
    G = S // patch
    cols = torch.empty((B * G * G, Cin * patch * patch), dtype=dtype, device=img.device)
    L.check(L.lib().mvlt_im2col_patch(_DT[dtype], _p(img), _p(cols), B, Cin, S, patch, _stream()), "mvlt_im2col_patch")
    return cols


def pack_plan(text_ids, labels, n_img):
    """Packing plan of a ragged caption batch, computed on the device (mvlt_pack_plan, no host sync).
    -> (row_start i32 [B], seq_len i32 [B], total i32 [1], row_start i64 [B], text_row i64 [B*T])"""
    _need_cuda(text_ids)
    assert text_ids.dtype == torch.int64 and text_ids.is_contiguous() and text_ids.dim() == 2
    B, T = text_ids.shape
    if labels is not None:
        assert labels.dtype == torch.int64 and labels.is_contiguous() and labels.numel() == B * T
    dev = text_ids.device
    i32buf = torch.empty(2 * B + 1, dtype=torch.int32, device=dev)
    i64buf = torch.empty(B + B * T, dtype=torch.int64, device=dev)
    rs, sl, tot = i32buf[:B], i32buf[B:2 * B], i32buf[2 * B:]
    rs64, trow = i64buf[:B], i64buf[B:]
    L.check(L.lib().mvlt_pack_plan(_p(text_ids), _p(labels), B, T, n_img, _p(rs), _p(sl), _p(tot), _p(rs64), _p(trow),
                                   _stream()), "mvlt_pack_plan")
    return rs, sl, tot, rs64, trow


def label_plan(labels, text_row):
    """Labelled caption positions first (mvlt_label_plan) -> (gather_row i32 [N], sel_labels i64 [N], count i32 [1])."""
    _need_cuda(labels)
    assert labels.dtype == torch.int64 and labels.is_contiguous()
    N = labels.numel()
    dev = labels.device
    gr = torch.empty(N + 1, dtype=torch.int32, device=dev)
    sel = torch.empty(N, dtype=torch.int64, device=dev)
    L.check(L.lib().mvlt_label_plan(_p(labels), _p(text_row), N, _p(gr), _p(sel), C.c_void_p(gr.data_ptr() + 4 * N), _stream()),
            "mvlt_label_plan")
    return gr[:N], sel, gr[N:]


def rows_scatter(x, rowmap, count, out):
    """out[rowmap[i]] = x[i] for i < count (device scalar)."""
    L.check(L.lib().mvlt_rows_scatter(_dt(x), _p(x), _p(out), x.shape[0], x.shape[1], _p(rowmap), _p(count), _stream()),
            "mvlt_rows_scatter")
    return out


def _embed_struct(dtype, B, n_img, T, H, text_ids, word, pos, typ, cls_id, sep_id, pos_offset, type_override):
    p = L.MvltEmbed()
    p.dtype, p.B, p.n_img, p.T, p.H = _DT[dtype], B, n_img, T, H
    if text_ids is not None:
        assert text_ids.dtype == torch.int64 and text_ids.is_contiguous()
        p.text_ids = _p(text_ids)
    p.word_emb, p.pos_emb, p.type_emb = _p(word), _p(pos), _p(typ)
    if torch.is_tensor(pos_offset):          # position kept on the device (replayed decode step)
        assert pos_offset.dtype == torch.int32 and pos_offset.is_cuda
        p.pos_offset, p.pos_offset_dev = 0, _p(pos_offset)
    else:
        p.pos_offset = pos_offset
    p.cls_id, p.sep_id, p.type_override = cls_id, sep_id, type_override
    return p


def embed_fwd(text_ids, image_feature, word, pos, typ, cls_id, sep_id, *, dtype=None, pos_offset=0,
              type_override=-1, pack=None, out=None):
    """MVLBert.get_embedding sum.  image_feature None -> cached-step (text only) layout."""
    if image_feature is not None:
        B, n_img, H = image_feature.shape
        dtype = image_feature.dtype
        assert image_feature.is_contiguous()
    else:
        B, n_img, H = text_ids.shape[0], -1, word.shape[1]
    T = 0 if text_ids is None else text_ids.shape[1]
    Lq = T if n_img < 0 else n_img + 2 + T
    if out is None:
        out = torch.empty((B, Lq, H) if pack is None else (pack[2], H), dtype=dtype, device=word.device)
    p = _embed_struct(dtype, B, n_img, T, H, text_ids, word, pos, typ, cls_id, sep_id, pos_offset, type_override)
    p.image_feature, p.out = _p(image_feature), _p(out)
    if pack is not None:
        p.row_start, p.seq_len = _p(pack[0]), _p(pack[1])
    L.check(L.lib().mvlt_embed_fwd(C.byref(p), _stream()), "mvlt_embed_fwd")
    return out


def embed_bwd(dout, text_ids, n_img, word, pos, typ, cls_id, sep_id, dword, dpos, dtype_emb, *, want_dimage=True,
              pos_offset=0, type_override=-1, pack=None, B=None):
    if pack is None:
        B, Lq, H = dout.shape
    else:
        H = dout.shape[1]            # dout: packed [R, H]; B given by the caller
    T = 0 if text_ids is None else text_ids.shape[1]
    dimg = torch.empty((B, n_img, H), dtype=dout.dtype, device=dout.device) if (want_dimage and n_img > 0) else None
    p = _embed_struct(dout.dtype, B, n_img, T, H, text_ids, word, pos, typ, cls_id, sep_id, pos_offset, type_override)
    assert dout.is_contiguous()
    p.dout, p.dimage, p.dword, p.dpos, p.dtype_emb = _p(dout), _p(dimg), _p(dword), _p(dpos), _p(dtype_emb)
    # the position / type table gradients are OVERWRITTEN with ordered batch sums and their unused rows zeroed (no atomics);
    # dword is accumulated: the caller clears it
    p.pos_rows = 0 if dpos is None else dpos.shape[0]
    p.type_rows = 0 if dtype_emb is None else dtype_emb.shape[0]
    if pack is not None:
        p.row_start, p.seq_len = _p(pack[0]), _p(pack[1])
    L.check(L.lib().mvlt_embed_bwd(C.byref(p), _stream()), "mvlt_embed_bwd")
    return dimg


def rows_transform(x, *, rowmap=None, rowscale=None, dropout=None, out=None):
    """out[i] = scale * mask(x[rowmap[i]]) (DropPath / dropout backward, window gathers)."""
    _need_cuda(x)
    assert x.is_contiguous() and x.dim() == 2
    rows = x.shape[0] if rowmap is None else rowmap.numel()
    if out is None:
        out = torch.empty((rows, x.shape[1]), dtype=x.dtype, device=x.device)
    dp, seed, tag = (dropout if dropout is not None else (0.0, 0, 0))
    rs, rps = (rowscale if rowscale is not None else (None, 1))
    L.check(L.lib().mvlt_rows_transform(_dt(x), _p(x), _p(out), rows, x.shape[1], _p(rowmap), _p(rs), rps,
                                        float(dp), int(seed), int(tag), _stream()), "mvlt_rows_transform")
    return out


def cast(x, dtype, out=None):
    _need_cuda(x)
    assert x.is_contiguous()
    if out is None:
        out = torch.empty(x.shape, dtype=dtype, device=x.device)
    L.check(L.lib().mvlt_cast(_dt(x), _p(x), _DT[dtype], _p(out), x.numel(), _stream()), "mvlt_cast")
    return out


def gelu(x):
    y = torch.empty_like(x)
    L.check(L.lib().mvlt_gelu_fwd(_dt(x), _p(x), _p(y), x.numel(), _stream()), "mvlt_gelu_fwd")
    return y


def tanh_fwd(x):
    y = torch.empty_like(x)
    L.check(L.lib().mvlt_tanh_fwd(_dt(x), _p(x), _p(y), x.numel(), _stream()), "mvlt_tanh_fwd")
    return y


def tanh_bwd(y, dy):
    dx = torch.empty_like(y)
    L.check(L.lib().mvlt_tanh_bwd(_dt(y), _p(y), _p(dy), _p(dx), y.numel(), _stream()), "mvlt_tanh_bwd")
    return dx


def dropout_mask(n, p, seed, tag, device):
    keep = torch.empty(n, dtype=torch.uint8, device=device)
    L.check(L.lib().mvlt_dropout_mask(_p(keep), n, float(p), int(seed), int(tag), _stream()), "mvlt_dropout_mask")
    return keep


def droppath_scale(B, p, seed, tag, device):
    s = torch.empty(B, dtype=torch.float32, device=device)
    L.check(L.lib().mvlt_droppath_scale(_p(s), B, float(p), int(seed), int(tag), _stream()), "mvlt_droppath_scale")
    return s


def droppath_scales(probs, B, seed, tag=0):
    """probs: f32 [rows] on the device -> [rows, B] scales (1/(1-p) kept, 0 dropped), one launch for a whole forward pass."""
    _need_cuda(probs)
    assert probs.dtype == torch.float32 and probs.is_contiguous()
    s = torch.empty((probs.numel(), B), dtype=torch.float32, device=probs.device)
    L.check(L.lib().mvlt_droppath_scales(_p(s), _p(probs), probs.numel(), B, int(seed), int(tag), _stream()), "mvlt_droppath_scales")
    return s


# ----------------------------------------------------------------------------- loss / optimizer / decode
def ce_fwd(logits, V, labels, rows_dev=None):
    """logits [rows, ld>=V]; returns (loss_sum, count, lse) device tensors (mean = sum/count).
    rows_dev (int32 device scalar): only the first *rows_dev rows exist."""
    _need_cuda(logits)
    rows = logits.shape[0]
    assert labels.dtype == torch.int64 and labels.numel() == rows and labels.is_contiguous()
    acc = torch.zeros(2, dtype=torch.float32, device=logits.device)
    lse = torch.empty(rows, dtype=torch.float32, device=logits.device)
    L.check(L.lib().mvlt_ce_fwd_ragged(_dt(logits), _p(logits), logits.stride(0), rows, V, _p(labels), _p(lse),
                                       C.c_void_p(acc.data_ptr()), C.c_void_p(acc.data_ptr() + 4), _p(rows_dev), _stream()),
            "mvlt_ce_fwd")
    return acc, lse


def ce_bwd(logits, V, labels, lse, acc, grad_scale=1.0, out=None, grad_scale_dev=None, rows_dev=None):
    if out is None:
        out = logits
    if grad_scale_dev is not None:
        assert grad_scale_dev.dtype == torch.float32 and grad_scale_dev.numel() == 1
    L.check(L.lib().mvlt_ce_bwd_ragged(_dt(logits), _p(logits), logits.stride(0), logits.shape[0], V, _p(labels), _p(lse),
                                       C.c_void_p(acc.data_ptr() + 4), float(grad_scale), _p(grad_scale_dev), _p(out),
                                       _p(rows_dev), _stream()), "mvlt_ce_bwd")
    return out


def gelu_bwd(x, dy, out=None):
    dx = torch.empty_like(x) if out is None else out
    L.check(L.lib().mvlt_gelu_bwd(_dt(x), _p(x), _p(dy), _p(dx), x.numel(), _stream()), "mvlt_gelu_bwd")
    return dx


def softmax_rows(logits, V):
    out = torch.empty((logits.shape[0], V), dtype=torch.float32, device=logits.device)
    L.check(L.lib().mvlt_softmax_rows(_dt(logits), _p(logits), logits.stride(0), logits.shape[0], V, _p(out),
                                      _stream()), "mvlt_softmax_rows")
    return out


def adamw(param, grad, exp_avg, exp_avg_sq, shadow, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    _need_cuda(param)
    L.check(L.lib().mvlt_adamw(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), _p(shadow), param.numel(),
                               float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step),
                               float(grad_scale), _stream()), "mvlt_adamw")


def attn_cached(qkv_new, k_cache, v_cache, past, scale, out=None):
    """past: int, or an int32 device tensor [1] (position kept on the GPU by a replayed decode loop)."""
    B, nH, cap, hd = k_cache.shape
    n_new = qkv_new.shape[0] // B
    if out is None:
        out = torch.empty((B * n_new, nH * hd), dtype=qkv_new.dtype, device=qkv_new.device)
    p = L.MvltAttnCached()
    p.dtype, p.B, p.nH, p.hd, p.n_new, p.cache_cap = _dt(qkv_new), B, nH, hd, n_new, cap
    if torch.is_tensor(past):
        assert past.dtype == torch.int32 and past.is_cuda
        p.past_dev = _p(past)
    else:
        p.past = past
    p.qkv_new, p.k_cache, p.v_cache, p.out, p.scale = _p(qkv_new), _p(k_cache), _p(v_cache), _p(out), float(scale)
    L.check(L.lib().mvlt_attn_cached(C.byref(p), _stream()), "mvlt_attn_cached")
    return out


def argmax(logits, V):
    out = torch.empty(logits.shape[0], dtype=torch.int64, device=logits.device)
    L.check(L.lib().mvlt_argmax(_dt(logits), _p(logits), logits.stride(0), logits.shape[0], V, _p(out), _stream()),
            "mvlt_argmax")
    return out
