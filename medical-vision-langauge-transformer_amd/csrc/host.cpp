// Native host path (torch C++ extension `_mvlt_host`) over the C-ABI of include/mvlt_hip.h.
//
// The Python modules (swin.py / bert.py) keep the parameters and decide WHAT runs; this file issues the launch
// sequence of one SwinTransformerBlock / BertLayer forward or backward pass in ONE native call: parameter-struct
// fills, activation allocation (torch caching allocator through ATen), the side-stream fork for the weight
// gradients and the deferred LayerNorm parameter reductions all happen here.  One step = ~800 kernel launches;
// through Python + ctypes each launch costs ~16 us of host time (13 ms per step), here ~4 us.
//
// Reference arithmetic: SwinTransformerBlock.forward (visual_feature_extractor.py:350-387), WindowAttention.forward
// (:224-254), Mlp.forward (:135-141); HF BertLayer (modeling_bert.py:111-136,164-203,282-293,325-351).
// torch is plumbing only: device memory and stream handles.
#include <torch/extension.h>
#include <hip/hip_runtime_api.h>
#include <vector>
#include <tuple>
#include <cstdlib>
#include <string>
#include "../../include/mvlt_hip.h"

namespace {

using at::Tensor;
using Ptrs = std::vector<int64_t>;

// The scratch buffers, the deferred LayerNorm queue, the event pool and the keep-alive list below are process-global, i.e.
// they belong to ONE device (one process per GPU): a second device in the same process would share
// them.  Every native entry point checks that it is still on the device the first call ran on.
int g_device = -1;
inline void check_device(const at::Tensor& t) {
    const int d = (int)t.get_device();
    if (g_device < 0) g_device = d;
    TORCH_CHECK(d == g_device, "mvlt_amd: the native host path holds per-process scratch for cuda:", g_device,
                " but was called with a tensor on cuda:", d, " (one process per GPU; MVLT_NATIVE_HOST=0 has no such state)");
}

inline void ck(int rc, const char* what) { TORCH_CHECK(rc == MVLT_OK, "mvlt_amd: ", what, " failed with status ", rc); }
inline int dtype_of(const Tensor& t) {
    if (t.scalar_type() == at::kFloat) return MVLT_F32;
    TORCH_CHECK(t.scalar_type() == at::kBFloat16, "mvlt_amd: float32 / bfloat16 only");
    return MVLT_BF16;
}
template <typename T = void> inline T* P(int64_t v) { return reinterpret_cast<T*>(v); }
inline void* dp(const Tensor& t) { return t.data_ptr(); }
inline float* fp(const Tensor& t) { return static_cast<float*>(t.data_ptr()); }
inline Tensor empty2(int64_t r, int64_t c, const Tensor& like) { return at::empty({r, c}, like.options()); }
inline Tensor emptyf(at::IntArrayRef sz, const Tensor& like) { return at::empty(sz, like.options().dtype(at::kFloat)); }

// ----------------------------------------------------------------------------------------------- streams
struct Streams { void* main = nullptr; void* side = nullptr; };
std::vector<hipEvent_t> g_events;
size_t g_event_next = 0;
std::vector<Tensor> g_side_keepalive;

hipEvent_t next_event() {
    if (g_events.empty()) {
        g_events.resize(64);
        for (auto& e : g_events) TORCH_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess, "hipEventCreate");
    }
    hipEvent_t e = g_events[g_event_next];
    g_event_next = (g_event_next + 1) % g_events.size();
    return e;
}
// experiments: MVLT_SIDE=0 issues the weight gradients on the main stream (no overlap with the dgrad chain)
// Fork behind the attention backward of a layer: its gradient is the last operand the layer's weight gradients need (the
// products behind it only feed the dgrad chain), and the event completes with that kernel itself (mvlt_attn_bwd_ev: no marker
// packet on the main stream, ~2.5 instead of ~5 us of main-stream time per layer; scripts/event_cost.hip).
void attn_bwd_fork(const MvltAttn& p, Streams& s) {
    hipEvent_t e = next_event();
    ck(mvlt_attn_bwd_ev(&p, s.main, e), "mvlt_attn_bwd_ev");
    TORCH_CHECK(hipStreamWaitEvent(static_cast<hipStream_t>(s.side), e, 0) == hipSuccess, "hipStreamWaitEvent");
}


// ----------------------------------------------------------------------------------------------- workspaces
struct Scratch { Tensor buf; std::vector<Tensor> retired; };
Scratch g_ws_main, g_ws_side;
void* workspace(Scratch& s, size_t need, const Tensor& like) {
    if (!s.buf.defined() || (size_t)s.buf.numel() < need) {
        if (s.buf.defined()) s.retired.push_back(s.buf);          // queued kernels may still use it
        s.buf = at::empty({(int64_t)std::max<size_t>(need, (size_t)64 << 20)}, like.options().dtype(at::kByte));
    }
    return s.buf.data_ptr();
}

// optional timing of sampled launches (bench.py roofline): HIP events on the launch stream around 1 in `every` launches
// of a kernel family.  FLOPs are counted with the row count the kernels EXECUTE: ragged batches keep it on the device
// (MvltGemm.m_dev), so the sample copies it to a pinned host slot on the same stream (no sync) and the FLOPs are
// evaluated in timer_collect -- counting the dense upper bound overstated the round-2 roofline by 13 %.
struct KernelTimer {
    bool on = false; int every = 4, count = 0;
    bool subtract_record_cost = true;     // checked against the kernel trace of the same run (scripts/roofline_vs_trace.py): the
                                          // main-stream family reads +27 % without and +0.2 % with it; the grouped launches on
                                          // the side stream, which start behind a cross-stream wait, +2 % without and -5 % with
    std::vector<hipEvent_t> ev;
    std::vector<double> mn2;          // per sample: 2 * sum_i (the two dimensions that are not ragged)
    std::vector<double> bfix, brow;   // per sample: ALGORITHMIC bytes = bfix + rows * brow (operands read once, outputs written once)
    std::vector<int> bound;           // per sample: host-known upper bound of the ragged dimension
    int* dev_rows = nullptr;          // pinned host: ragged dimension read from the device (-1: none, use the bound)
    size_t used = 0, cap = 0;
    bool sample() { return on && (count++ % every) == 0 && used < cap; }
    void begin(void* stream, double mn2_, int bound_, const int32_t* rows_dev, double bfix_ = 0.0, double brow_ = 0.0) {
        mn2.push_back(mn2_); bound.push_back(bound_); bfix.push_back(bfix_); brow.push_back(brow_);
        dev_rows[used] = -1;
        if (rows_dev) (void)hipMemcpyAsync(&dev_rows[used], rows_dev, sizeof(int), hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream));
        (void)hipEventRecord(ev[3 * used], static_cast<hipStream_t>(stream));
    }
    // a third event straight behind the second: (e2 - e1) is what one event record costs on this stream at this point of
    // the step, and collect() takes it off (e1 - e0) -- without that the bracket of a 25 us kernel reads 27 % high
    // against the profiler's kernel duration (profiles/README.md, round 3)
    void end(void* stream) {
        (void)hipEventRecord(ev[3 * used + 1], static_cast<hipStream_t>(stream));
        (void)hipEventRecord(ev[3 * used + 2], static_cast<hipStream_t>(stream));
        ++used;
    }
    void reset(int every_, size_t cap_) {
        for (auto e : ev) (void)hipEventDestroy(e);
        if (dev_rows) (void)hipHostFree(dev_rows);
        ev.clear(); mn2.clear(); bound.clear(); bfix.clear(); brow.clear(); dev_rows = nullptr; used = 0; count = 0;
        on = cap_ > 0; every = std::max(1, every_); cap = cap_;
        if (!on) return;
        ev.resize(3 * cap);
        for (auto& e : ev) TORCH_CHECK(hipEventCreate(&e) == hipSuccess, "hipEventCreate");
        TORCH_CHECK(hipHostMalloc(reinterpret_cast<void**>(&dev_rows), cap * sizeof(int), hipHostMallocDefault) == hipSuccess, "hipHostMalloc");
    }
    // [(executed flops, milliseconds, algorithmic bytes)]; call after a device synchronize
    std::vector<std::tuple<double, double, double>> collect() {
        std::vector<std::tuple<double, double, double>> out;
        for (size_t i = 0; i < used; ++i) {
            float ms = 0.f, over = 0.f;
            if (hipEventElapsedTime(&ms, ev[3 * i], ev[3 * i + 1]) != hipSuccess) continue;
            if (subtract_record_cost && hipEventElapsedTime(&over, ev[3 * i + 1], ev[3 * i + 2]) == hipSuccess && over < ms) ms -= over;
            const int rows = dev_rows[i] >= 0 ? std::min(dev_rows[i], bound[i]) : bound[i];
            out.emplace_back(mn2[i] * rows, (double)ms, bfix[i] + brow[i] * rows);
        }
        on = false;
        return out;
    }
};
KernelTimer g_timer = [] { KernelTimer t; t.subtract_record_cost = false; return t; }();        // grouped weight-gradient launches (gemm_group_kernel)
KernelTimer g_timer_fam;    // forward / dgrad products of the layers (gemm_kernel, gemm_glds_kernel, gemm8_kernel)

// ----------------------------------------------------------------------------------------------- GEMM
struct Epi {
    const void* pf = nullptr; int64_t pf_bytes = 0;      // MvltGemm.prefetch: the weights of the product that runs next
    const float* bias = nullptr; bool gelu = false; void* pre = nullptr;
    float drop_p = 0.f; uint64_t seed = 0; uint32_t tag = 0;
    const float* rowscale = nullptr; int rps = 1;
    const void* residual = nullptr; int64_t ldr = 0;
    const int32_t* rowmap = nullptr; const void* aux = nullptr;
    bool out_f32 = false, accumulate = false; float* a_colsum = nullptr;
    const int32_t* m_dev = nullptr;          // valid storage rows of A on the device (ragged batches)
};

void fill_gemm(MvltGemm& p, int dtype, int M, int N, int K, const void* A, int64_t lda, bool ak, const void* B, int64_t ldb,
               bool bk, void* C, int64_t ldc, const Epi& e) {
    p = MvltGemm{};
    p.dtype = dtype; p.M = M; p.N = N; p.K = K;
    p.A = A; p.lda = lda; p.a_kmajor = ak; p.B = B; p.ldb = ldb; p.b_kmajor = bk; p.C = C; p.ldc = ldc;
    int epi = 0;
    if (e.bias) { epi |= MVLT_EPI_BIAS; p.bias = e.bias; }
    if (e.gelu) { epi |= MVLT_EPI_GELU; if (e.pre) { epi |= MVLT_EPI_SAVE_PRE; p.pre = e.pre; } }
    if (e.drop_p > 0.f) { epi |= MVLT_EPI_DROPOUT; p.dropout_p = e.drop_p; p.seed = e.seed; p.tag = e.tag; }
    if (e.rowscale) { epi |= MVLT_EPI_ROWSCALE; p.rowscale = e.rowscale; p.rows_per_scale = e.rps; }
    if (e.residual) { epi |= MVLT_EPI_RESIDUAL; p.residual = e.residual; p.ldr = e.ldr; }
    if (e.rowmap) { epi |= MVLT_EPI_ROWMAP; p.rowmap = e.rowmap; }
    if (e.aux) { epi |= MVLT_EPI_MUL_GELU_GRAD; p.aux = e.aux; }
    if (e.out_f32) epi |= MVLT_EPI_OUT_F32;
    if (e.accumulate) epi |= MVLT_EPI_ACCUM;
    p.epilogue = epi;
    p.a_colsum = e.a_colsum;
    p.m_dev = e.m_dev;
    p.prefetch = e.pf; p.prefetch_bytes = e.pf_bytes;          // the next product's weights, pulled towards the caches (15.3 -> 15.0 ms per step, round 2)
}

void gemm(int dtype, int M, int N, int K, const void* A, int64_t lda, bool ak, const void* B, int64_t ldb, bool bk,
          void* C, int64_t ldc, const Epi& e, void* stream, Scratch& ws, const Tensor& like) {
    MvltGemm p;
    fill_gemm(p, dtype, M, N, K, A, lda, ak, B, ldb, bk, C, ldc, e);
    const size_t need = mvlt_gemm_workspace_bytes(&p);
    if (need) { p.workspace = workspace(ws, need, like); p.workspace_bytes = (size_t)ws.buf.numel(); }
    const bool timed = !ak && g_timer_fam.sample();          // forward / dgrad products: rows (M) may be ragged
    if (timed) {
        // algorithmic bytes of the product: the weight once, per row the activation row in and the output row out,
        // plus what the fused epilogue reads / writes per row (residual, saved pre-activation, gelu' operand)
        const double esz = dtype == MVLT_BF16 ? 2.0 : 4.0;
        const double per_row = (K + N * (e.out_f32 ? 4.0 / esz : 1.0) + (e.residual ? N : 0) + (e.pre ? N : 0) + (e.aux ? N : 0)) * esz;
        g_timer_fam.begin(stream, 2.0 * N * K, M, e.m_dev, (double)N * K * esz, per_row);
    }
    ck(mvlt_gemm(&p, stream), "mvlt_gemm");
    if (timed) g_timer_fam.end(stream);
}

// y = x W^T (+ epilogue): x [rows, K] contiguous, W [N, K]
inline void linear(const Tensor& x, int64_t W, int N, Tensor& out, const Epi& e, void* stream) {
    gemm(dtype_of(x), (int)x.size(0), N, (int)x.size(1), dp(x), x.size(1), false, P(W), x.size(1), false, dp(out), out.size(1),
         e, stream, g_ws_main, x);
}
// dx = dy W (+ epilogue): dy [rows, N], W [N, K] read k-major
inline void dgrad(const Tensor& dy, int64_t W, int K, Tensor& out, const Epi& e, void* stream) {
    gemm(dtype_of(dy), (int)dy.size(0), K, (int)dy.size(1), dp(dy), dy.size(1), false, P(W), K, true, dp(out), out.size(1),
         e, stream, g_ws_main, dy);
}

struct WItem { const Tensor* dy; const Tensor* x; int64_t dw; int64_t db; };
// weight gradients of one layer: dW_i = dY_i^T X_i, db_i = colsum(dY_i)  (ops.wgrad_group)
void wgrad_group(const std::vector<WItem>& items, void* stream, Scratch& ws, const int32_t* k_dev = nullptr) {
    const int n = (int)items.size();
    bool all128 = true, all96 = true;
    for (auto& it : items) { all128 &= it.x->size(1) % 128 == 0; all96 &= it.x->size(1) % 96 == 0; }
    const int bn = all128 ? 128 : (all96 ? 96 : 0);
    long tiles = 0;
    if (bn) for (auto& it : items) tiles += ((it.dy->size(1) + 63) / 64) * ((it.x->size(1) + bn - 1) / bn);
    const int dtype = dtype_of(*items[0].dy);
    constexpr long long_k = 8192;          // reduction rows from which a group with few tiles is cut into k-slices (Swin stages 0 / 1)
    // few tiles but a long reduction (Swin stages 0/1): still one launch, cut into k-slices inside mvlt_gemm_group (bf16: the
    // 8-wave engine's slab reduce, deterministic; the exact-f32 parity mode takes split-K slabs + the reduce kernel, one
    // product at a time)
    const bool slices_ok = dtype == MVLT_BF16;
    if (!(n > 1 && n <= 8 && bn && (tiles >= 200 || (slices_ok && items[0].dy->size(0) >= long_k)))) {
        for (auto& it : items) {
            Epi e; e.out_f32 = true; e.a_colsum = P<float>(it.db); e.m_dev = k_dev;
            gemm(dtype, (int)it.dy->size(1), (int)it.x->size(1), (int)it.dy->size(0), dp(*it.dy), it.dy->size(1), true,
                 dp(*it.x), it.x->size(1), true, P(it.dw), it.x->size(1), e, stream, ws, *it.dy);
        }
        return;
    }
    MvltGemm arr[8];
    double mn2 = 0, bfix = 0, brow = 0;        // algorithmic bytes: per reduction row both operand rows (bf16), once the f32 dW + db
    for (int i = 0; i < n; ++i) {
        const auto& it = items[i];
        Epi e; e.out_f32 = true; e.a_colsum = P<float>(it.db); e.m_dev = k_dev;
        fill_gemm(arr[i], dtype, (int)it.dy->size(1), (int)it.x->size(1), (int)it.dy->size(0), dp(*it.dy), it.dy->size(1), true,
                  dp(*it.x), it.x->size(1), true, P(it.dw), it.x->size(1), e);
        arr[i].split_k = 1;
        mn2 += 2.0 * arr[i].M * arr[i].N;          // the reduction length (activation rows) is the ragged dimension here
        brow += (double)(arr[i].M + arr[i].N) * it.dy->element_size();
        bfix += ((double)arr[i].M * arr[i].N + arr[i].M) * 4.0;
    }
    const size_t need = mvlt_gemm_group_workspace_bytes(arr, n);          // k-slice slabs of the 8-wave engine
    if (need) { arr[0].workspace = workspace(ws, need, *items[0].dy); arr[0].workspace_bytes = (size_t)ws.buf.numel(); }
    const bool timed = g_timer.sample();
    if (timed) g_timer.begin(stream, mn2, arr[0].K, k_dev, bfix, brow);
    ck(mvlt_gemm_group(arr, n, stream), "mvlt_gemm_group");
    if (timed) g_timer.end(stream);
}

// ----------------------------------------------------------------------------------------------- LayerNorm
void ln_fwd(const Tensor& x, int rows, int C, int64_t gamma, int64_t beta, float eps, Tensor& y, float* mean, float* rstd,
            const int32_t* out_rowmap, void* stream, const int32_t* rows_dev = nullptr) {
    MvltLayerNorm p{};
    p.dtype = dtype_of(x); p.rows = rows; p.C = C; p.eps = eps;
    p.x = dp(x); p.gamma = P<float>(gamma); p.beta = P<float>(beta); p.y = dp(y);
    p.mean = mean; p.rstd = rstd; p.out_rowmap = out_rowmap; p.rows_dev = rows_dev;
    ck(mvlt_layernorm_fwd(&p, stream), "mvlt_layernorm_fwd");
}

// deferred dgamma/dbeta reductions of one backward pass (ops.LnReduceQueue)
struct LnQueue {
    Tensor pool; std::vector<Tensor> retired; int64_t off = 0;
    std::vector<MvltLnReduceItem> items;
    float* take(int64_t nfloats, const Tensor& like) {
        if (!pool.defined() || pool.numel() < off + nfloats) {
            if (pool.defined()) retired.push_back(pool);
            pool = at::empty({std::max<int64_t>(off + nfloats, (int64_t)48 << 20)}, like.options().dtype(at::kFloat));
            off = 0;
        }
        float* w = fp(pool) + off;
        off += nfloats;
        return w;
    }
    std::vector<MvltSwinDbiasItem> dbias;          // per-workgroup bias-gradient sums of mvlt_swin_wmsa2_bwd launches (same pool)
    void flush(void* stream) {
        if (!dbias.empty()) {
            ck(mvlt_swin_wmsa2_bwd_dbias(dbias.data(), (int)dbias.size(), stream), "mvlt_swin_wmsa2_bwd_dbias");
            dbias.clear();
        }
        if (items.empty()) { off = 0; retired.clear(); return; }
        ck(mvlt_layernorm_param_reduce_batch(items.data(), (int)items.size(), stream), "mvlt_layernorm_param_reduce_batch");
        items.clear(); off = 0; retired.clear();
    }
};
LnQueue g_lnq;

struct LnBranch { void* dz = nullptr; const int32_t* rowmap = nullptr; const float* rowscale = nullptr; int rps = 1;
                  float drop_p = 0.f; uint64_t seed = 0; uint32_t tag = 0; };
void ln_bwd(const Tensor& dy, const int32_t* dy_rowmap, const Tensor& x, const float* mean, const float* rstd, int rows, int C,
            int64_t gamma, int64_t dgamma, int64_t dbeta, const void* dres, Tensor& dx, const LnBranch& br, void* stream,
            const int32_t* rows_dev = nullptr, int dy_parts = 0) {
    static const int ws_rows = mvlt_layernorm_bwd_workspace_rows();
    MvltLayerNormBwd p{};
    p.dtype = dtype_of(x); p.rows = rows; p.C = C;
    if (dy_parts > 1) { p.dy_parts = dy_parts; p.dy_part_stride = (int64_t)rows * C; }          // dy: [dy_parts][rows, C]
    p.dy = dp(dy); p.dy_rowmap = dy_rowmap; p.x = dp(x); p.mean = mean; p.rstd = rstd; p.gamma = P<float>(gamma);
    p.dres = dres; p.dx = dp(dx);
    p.dgamma = P<float>(dgamma); p.dbeta = P<float>(dbeta);
    float* ws = g_lnq.take((int64_t)2 * ws_rows * C, x);
    p.workspace = ws; p.defer_param_reduce = 1; p.rows_dev = rows_dev;
    g_lnq.items.push_back(MvltLnReduceItem{ws, mvlt_layernorm_bwd_nparts(rows, C), C, p.dgamma, p.dbeta});
    if (br.dz) {
        p.dz = br.dz; p.dz_rowmap = br.rowmap; p.dz_rowscale = br.rowscale; p.dz_rows_per_scale = br.rps;
        p.dz_dropout_p = br.drop_p; p.seed = br.seed; p.tag = br.tag;
    }
    ck(mvlt_layernorm_bwd(&p, stream), "mvlt_layernorm_bwd");
}

// ----------------------------------------------------------------------------------------------- BERT layer
// w: compute-dtype weights [wqkv, wo, wi, wo2]; f: f32 [bqkv, bo, bi, bo2, g1, b1, g2, b2];
// g: f32 gradients [dwqkv, dbqkv, dwo, dbo, dwi, dbi, dwo2, dbo2, dg1, db1, dg2, db2]
struct AttnArgs { int mode, B, Lq, nH; int64_t text_ids; int T; int64_t image_mask; int obj_end; int64_t row_start, seq_len; };

void fill_attn(MvltAttn& p, const Tensor& qkv, const AttnArgs& a, int H, void* out, float* lse, float drop_p, uint64_t seed, uint32_t tag) {
    p = MvltAttn{};
    p.dtype = dtype_of(qkv); p.mode = a.mode; p.nseq = a.B; p.L = a.Lq; p.nH = a.nH; p.hd = H / a.nH;
    p.qkv = dp(qkv); p.out = out; p.lse = lse; p.scale = 1.0f / std::sqrt((float)p.hd);
    p.text_ids = P<const int64_t>(a.text_ids); p.T = a.T; p.image_mask = P<const uint8_t>(a.image_mask); p.obj_end = a.obj_end;
    p.row_start = P<const int32_t>(a.row_start); p.seq_len = P<const int32_t>(a.seq_len);
    if (drop_p > 0.f) { p.dropout_p = drop_p; p.seed = seed; p.tag = tag; }
}

std::vector<Tensor> bert_layer_fwd(const Tensor& x, const Ptrs& w, const Ptrs& f, int64_t H, int64_t I, double eps,
                                   const Ptrs& attn, double p_h, double p_a, int64_t seed, int64_t layer, bool save,
                                   int64_t stream_) {
    check_device(x);
    void* st = P(stream_);
    const int64_t rows = x.size(0);
    AttnArgs a{(int)attn[0], (int)attn[1], (int)attn[2], (int)attn[3], attn[4], (int)attn[5], attn[6], (int)attn[7], attn[8], attn[9]};
    const uint64_t sd = (uint64_t)seed;
    const int32_t* rd = attn.size() > 10 ? P<const int32_t>(attn[10]) : nullptr;      // valid rows on the device (auto-packed batch)
    Tensor qkv = empty2(rows, 3 * H, x);
    const int64_t esz = x.element_size();
    const bool nb = w.size() >= 8;                       // w[4..7]: (ptr, bytes) of the next layer's qkv / the previous layer's FFN-out weights
    { Epi e; e.m_dev = rd; e.bias = P<float>(f[0]); e.pf = P(w[1]); e.pf_bytes = H * H * esz; linear(x, w[0], (int)(3 * H), qkv, e, st); }
    Tensor ctx = empty2(rows, H, x);
    Tensor lse = emptyf({a.B, a.nH, a.Lq}, x);
    { MvltAttn p; fill_attn(p, qkv, a, (int)H, dp(ctx), fp(lse), (float)p_a, sd, (uint32_t)(8 * layer + 0));
      ck(mvlt_attn_fwd(&p, st), "mvlt_attn_fwd"); }
    Tensor y1 = empty2(rows, H, x);
    { Epi e; e.m_dev = rd; e.bias = P<float>(f[1]); e.drop_p = (float)p_h; e.seed = sd; e.tag = (uint32_t)(8 * layer + 1);
      e.residual = dp(x); e.ldr = H; e.pf = P(w[2]); e.pf_bytes = I * H * esz; linear(ctx, w[1], (int)H, y1, e, st); }
    Tensor x1 = empty2(rows, H, x), st1, st2;
    float *m1 = nullptr, *r1 = nullptr, *m2 = nullptr, *r2 = nullptr;
    if (save) { st1 = emptyf({2, rows}, x); m1 = fp(st1); r1 = m1 + rows; st2 = emptyf({2, rows}, x); m2 = fp(st2); r2 = m2 + rows; }
    ln_fwd(y1, (int)rows, (int)H, f[4], f[5], (float)eps, x1, m1, r1, nullptr, st, rd);
    Tensor h = empty2(rows, I, x), act = empty2(rows, I, x);
    { Epi e; e.m_dev = rd; e.bias = P<float>(f[2]); e.gelu = true; e.pre = dp(h); e.pf = P(w[3]); e.pf_bytes = H * I * esz; linear(x1, w[2], (int)I, act, e, st); }
    Tensor y2 = empty2(rows, H, x);
    { Epi e; e.m_dev = rd; e.bias = P<float>(f[3]); e.drop_p = (float)p_h; e.seed = sd; e.tag = (uint32_t)(8 * layer + 2);
      e.residual = dp(x1); e.ldr = H; if (nb) { e.pf = P(w[4]); e.pf_bytes = w[5]; } linear(act, w[3], (int)H, y2, e, st); }
    Tensor x2 = empty2(rows, H, x);
    ln_fwd(y2, (int)rows, (int)H, f[6], f[7], (float)eps, x2, m2, r2, nullptr, st, rd);
    if (!save) return {x2};
    return {x2, x, qkv, ctx, lse, y1, st1, x1, h, act, y2, st2};
}

// saved = [x, qkv, ctx, lse, y1, st1, x1, h, act, y2, st2] (bert_layer_fwd output 1..)
Tensor bert_layer_bwd(const Tensor& dx, const std::vector<Tensor>& sv, const Ptrs& w, const Ptrs& f, const Ptrs& g, int64_t H,
                      int64_t I, const Ptrs& attn, double p_h, double p_a, int64_t seed, int64_t layer, int64_t stream_,
                      int64_t side_) {
    check_device(dx);
    Streams ss{P(stream_), P(side_)};
    void* st = ss.main;
    const Tensor &x = sv[0], &qkv = sv[1], &ctx = sv[2], &lse = sv[3], &y1 = sv[4], &st1 = sv[5], &x1 = sv[6], &h = sv[7],
                 &act = sv[8], &y2 = sv[9], &st2 = sv[10];
    const int64_t rows = x.size(0);
    AttnArgs a{(int)attn[0], (int)attn[1], (int)attn[2], (int)attn[3], attn[4], (int)attn[5], attn[6], (int)attn[7], attn[8], attn[9]};
    const uint64_t sd = (uint64_t)seed;
    const int32_t* rd = attn.size() > 10 ? P<const int32_t>(attn[10]) : nullptr;
    // LN2 backward: dy2 = gradient of y2 (also the residual branch into x1); dz2 = hidden-dropout backward of it
    Tensor dy2 = empty2(rows, H, x), dz2 = dy2;
    { LnBranch br; if (p_h > 0) { dz2 = empty2(rows, H, x); br.dz = dp(dz2); br.drop_p = (float)p_h; br.seed = sd; br.tag = (uint32_t)(8 * layer + 2); }
      ln_bwd(dx, nullptr, y2, fp(st2), fp(st2) + rows, (int)rows, (int)H, f[6], g[10], g[11], nullptr, dy2, br, st, rd); }
    Tensor dh = empty2(rows, I, x);
    const int64_t esz = x.element_size();
    const bool nb = w.size() >= 8;
    { Epi e; e.m_dev = rd; e.aux = dp(h); e.pf = P(w[2]); e.pf_bytes = I * H * esz; dgrad(dz2, w[3], (int)I, dh, e, st); }
    Tensor dx1 = empty2(rows, H, x);
    { Epi e; e.m_dev = rd; e.residual = dp(dy2); e.ldr = H; e.pf = P(w[1]); e.pf_bytes = H * H * esz; dgrad(dh, w[2], (int)H, dx1, e, st); }
    Tensor dy1 = empty2(rows, H, x), dz1 = dy1;
    { LnBranch br; if (p_h > 0) { dz1 = empty2(rows, H, x); br.dz = dp(dz1); br.drop_p = (float)p_h; br.seed = sd; br.tag = (uint32_t)(8 * layer + 1); }
      ln_bwd(dx1, nullptr, y1, fp(st1), fp(st1) + rows, (int)rows, (int)H, f[4], g[8], g[9], nullptr, dy1, br, st, rd); }
    Tensor dctx = empty2(rows, H, x);
    { Epi e; e.m_dev = rd; e.pf = P(w[0]); e.pf_bytes = 3 * H * H * esz; dgrad(dz1, w[1], (int)H, dctx, e, st); }
    Tensor dqkv = empty2(rows, 3 * H, x);
    { MvltAttn p; fill_attn(p, qkv, a, (int)H, dp(ctx), fp(lse), (float)p_a, sd, (uint32_t)(8 * layer + 0));
      Tensor delta = at::empty_like(lse);
      p.dout = dp(dctx); p.dqkv = dp(dqkv); p.delta_ws = fp(delta);
      attn_bwd_fork(p, ss); }
    // weight / bias gradients on the side stream (off the critical path)
    for (const Tensor* t : std::initializer_list<const Tensor*>{&dz2, &act, &dh, &x1, &dz1, &ctx, &dqkv, &x}) g_side_keepalive.push_back(*t);
    wgrad_group({{&dz2, &act, g[6], g[7]}, {&dh, &x1, g[4], g[5]}, {&dz1, &ctx, g[2], g[3]}, {&dqkv, &x, g[0], g[1]}},
                ss.side, g_ws_side, rd);
    Tensor dxin = empty2(rows, H, x);
    { Epi e; e.m_dev = rd; e.residual = dp(dy1); e.ldr = H; if (nb) { e.pf = P(w[6]); e.pf_bytes = w[7]; } dgrad(dqkv, w[0], (int)H, dxin, e, st); }
    return dxin;
}

// ----------------------------------------------------------------------------------------------- Swin block
// w: compute-dtype weights [wqkv, wproj, wfc1, wfc2]; f: f32 [n1g, n1b, bqkv, bproj, table, n2g, n2b, bfc1, bfc2];
// g: f32 gradients [dn1g, dn1b, dwqkv, dbqkv, dwproj, dbproj, dtable, dn2g, dn2b, dwfc1, dbfc1, dwfc2, dbfc2]
// geo: [B, H(res), C, nH, shift, fused (0 | 1 wmsa.hip | 2 wmsa2.hip), sync workspace of wmsa2]; maps: [w2n, n2w] int32 device pointers; s1/s2: DropPath scales f32 [B] or 0
std::vector<Tensor> swin_block_fwd(const Tensor& x, const Ptrs& w, const Ptrs& f, const Ptrs& geo, const Ptrs& maps, double scale,
                                   double eps, int64_t s1, int64_t s2, bool save, int64_t stream_) {
    check_device(x);
    void* st = P(stream_);
    const int B = (int)geo[0], res = (int)geo[1], C = (int)geo[2], nH = (int)geo[3], shift = (int)geo[4];
    const bool fused = geo[5] != 0;
    const bool fused2 = geo[5] == 2;                     // second design (csrc/wmsa2.hip): geo[6] = its int32 sync workspace
    const int64_t rows = x.size(0);
    const int Lt = res * res, nW = (res / 7) * (res / 7);
    const int dtype = dtype_of(x);
    const int64_t esz = x.element_size();
    const bool nb = w.size() >= 8;                       // w[4..7]: (ptr, bytes) of the next block's qkv / the previous block's fc2 weights
    Tensor xn1w, qkv, ao, lse, stat1, x1 = empty2(rows, C, x);
    float *m1 = nullptr, *r1 = nullptr;
    if (save) { stat1 = emptyf({2, rows}, x); m1 = fp(stat1); r1 = m1 + rows; }
    if (fused) {
        MvltSwinWmsa p{};
        p.dtype = dtype; p.B = B; p.res = res; p.C = C; p.nH = nH; p.shift = shift;
        p.x = dp(x); p.y = dp(x1); p.w2n = P<const int32_t>(maps[0]);
        p.ln_gamma = P<float>(f[0]); p.ln_beta = P<float>(f[1]); p.ln_eps = (float)eps;
        p.wqkv = P(w[0]); p.bqkv = P<float>(f[2]); p.wproj = P(w[1]); p.bproj = P<float>(f[3]);
        p.bias_table = P<float>(f[4]); p.scale = (float)scale; p.rowscale = P<float>(s1);
        if (save) {
            xn1w = empty2(rows, C, x); qkv = empty2(rows, 3 * C, x); ao = empty2(rows, C, x); lse = emptyf({rows / 49, nH, 49}, x);
            p.xn_win = dp(xn1w); p.qkv_win = dp(qkv); p.attn_out = dp(ao); p.lse = fp(lse); p.mean = m1; p.rstd = r1;
        }
        if (fused2) {
            if (!save) { ao = empty2(rows, C, x); p.attn_out = dp(ao); }       // the head groups meet through attn_out
            ck(mvlt_swin_wmsa2_fwd(&p, P<int32_t>(geo[6]), st), "mvlt_swin_wmsa2_fwd");
        } else {
            ck(mvlt_swin_wmsa_fwd(&p, st), "mvlt_swin_wmsa_fwd");
        }
    } else {
        xn1w = empty2(rows, C, x);
        ln_fwd(x, (int)rows, C, f[0], f[1], (float)eps, xn1w, m1, r1, P<const int32_t>(maps[1]), st);
        qkv = empty2(rows, 3 * C, x);
        { Epi e; e.bias = P<float>(f[2]); e.pf = P(w[1]); e.pf_bytes = (int64_t)C * C * esz; linear(xn1w, w[0], 3 * C, qkv, e, st); }
        ao = empty2(rows, C, x); lse = emptyf({rows / 49, nH, 49}, x);
        { MvltAttn p{}; p.dtype = dtype; p.mode = MVLT_ATTN_SWIN; p.nseq = B * nW; p.L = 49; p.nH = nH; p.hd = C / nH;
          p.qkv = dp(qkv); p.out = dp(ao); p.lse = fp(lse); p.scale = (float)scale;
          p.bias_table = P<float>(f[4]); p.nW = nW; p.win_res = res; p.shift = shift;
          ck(mvlt_attn_fwd(&p, st), "mvlt_attn_fwd"); }
        { Epi e; e.bias = P<float>(f[3]); e.residual = dp(x); e.ldr = C; e.rowmap = P<const int32_t>(maps[0]);
          if (s1) { e.rowscale = P<float>(s1); e.rps = Lt; }
          e.pf = P(w[2]); e.pf_bytes = (int64_t)4 * C * C * esz; linear(ao, w[1], C, x1, e, st); }
    }
    Tensor xn2 = empty2(rows, C, x), stat2;
    float *m2 = nullptr, *r2 = nullptr;
    if (save) { stat2 = emptyf({2, rows}, x); m2 = fp(stat2); r2 = m2 + rows; }
    ln_fwd(x1, (int)rows, C, f[5], f[6], (float)eps, xn2, m2, r2, nullptr, st);
    Tensor h = empty2(rows, 4 * C, x), act = empty2(rows, 4 * C, x);
    { Epi e; e.bias = P<float>(f[7]); e.gelu = true; e.pre = dp(h); e.pf = P(w[3]); e.pf_bytes = (int64_t)4 * C * C * esz; linear(xn2, w[2], 4 * C, act, e, st); }
    Tensor x2 = empty2(rows, C, x);
    { Epi e; e.bias = P<float>(f[8]); e.residual = dp(x1); e.ldr = C; if (s2) { e.rowscale = P<float>(s2); e.rps = Lt; }
      if (nb) { e.pf = P(w[4]); e.pf_bytes = w[5]; } linear(act, w[3], C, x2, e, st); }
    if (!save) return {x2};
    return {x2, x, stat1, xn1w, qkv, ao, lse, x1, stat2, xn2, h, act};
}

// sv = [x, stat1, xn1w, qkv, ao, lse, x1, stat2, xn2, h, act]
// s2_next: DropPath scales of the MLP branch of the block that consumes dx0 (the previous block of the stage), or 0: the
// last LayerNorm backward then also writes dx0 * s2_next, which that block takes as `dy2_pre` instead of launching a
// row-scale pass of its own.  Returns {dx0} or {dx0, dx0 * s2_next}.
std::vector<Tensor> swin_block_bwd(const Tensor& dx2, const std::vector<Tensor>& sv, const Ptrs& w, const Ptrs& f, const Ptrs& g, const Ptrs& geo,
                                   const Ptrs& maps, double scale, int64_t s1, int64_t s2, int64_t s2_next, const c10::optional<Tensor>& dy2_pre,
                                   int64_t stream_, int64_t side_) {
    check_device(dx2);
    Streams ss{P(stream_), P(side_)};
    void* st = ss.main;
    const Tensor &x = sv[0], &stat1 = sv[1], &xn1w = sv[2], &qkv = sv[3], &ao = sv[4], &lse = sv[5], &x1 = sv[6], &stat2 = sv[7],
                 &xn2 = sv[8], &h = sv[9], &act = sv[10];
    const int B = (int)geo[0], res = (int)geo[1], C = (int)geo[2], nH = (int)geo[3], shift = (int)geo[4];
    const int64_t rows = x.size(0);
    const int Lt = res * res, nW = (res / 7) * (res / 7);
    const int dtype = dtype_of(x);
    const int32_t* w2n = P<const int32_t>(maps[0]); const int32_t* n2w = P<const int32_t>(maps[1]);
    (void)w2n;
    Tensor dy2 = dx2;
    if (s2 && dy2_pre.has_value()) {
        dy2 = *dy2_pre;
        TORCH_CHECK(dy2.sizes() == dx2.sizes() && dy2.scalar_type() == dx2.scalar_type() && dy2.is_contiguous(), "dy2_pre");
    } else if (s2) {
        dy2 = empty2(rows, C, x);
        ck(mvlt_rows_transform(dtype, dp(dx2), dp(dy2), (int)rows, C, nullptr, P<float>(s2), Lt, 0.f, 0, 0, st), "mvlt_rows_transform");
    }
    Tensor dh = empty2(rows, 4 * C, x);
    const int64_t esz = x.element_size();
    const bool nb = w.size() >= 8;
    { Epi e; e.aux = dp(h); e.pf = P(w[2]); e.pf_bytes = (int64_t)4 * C * C * esz; dgrad(dy2, w[3], 4 * C, dh, e, st); }
    Tensor dxn2 = empty2(rows, C, x);
    { Epi e; e.pf = P(w[1]); e.pf_bytes = (int64_t)C * C * esz; dgrad(dh, w[2], C, dxn2, e, st); }
    Tensor dx1 = empty2(rows, C, x), dyw = empty2(rows, C, x);
    { LnBranch br; br.dz = dp(dyw); br.rowmap = n2w; if (s1) { br.rowscale = P<float>(s1); br.rps = Lt; }
      ln_bwd(dxn2, nullptr, x1, fp(stat2), fp(stat2) + rows, (int)rows, C, f[5], g[7], g[8], dp(dx2), dx1, br, st); }
    // the output projection's dgrad rides inside the attention backward where the kernel takes it (MvltAttn.dout_weight: bf16,
    // 3 / 6 / 12 heads of 32; MVLT_SWIN_BWD_PROJ=0: the separate product)
    static const bool proj_in_attn = [] { const char* e = getenv("MVLT_SWIN_BWD_PROJ"); return !e || e[0] != '0'; }();
    const bool fuse_proj = proj_in_attn && dtype == MVLT_BF16 && C == 32 * nH && (nH == 3 || nH == 6 || nH == 12) && (shift == 0 || shift == 3);
    // one launch for projection dgrad + attention backward + qkv dgrad where the second design covers the shape
    // (mvlt_swin_wmsa2_bwd: partial qkv-dgrad products per head group, summed by the LayerNorm backward below; MVLT_SWIN_BWD_ONE=0: off)
    // MVLT_SWIN_BWD_ONE = smallest width that takes it (default 384: stage 2; 96 / 192: stages 0 / 1 too, measured slower there --
    // a workgroup walks 4 / 2 units one after the other and every phase of a unit is latency-bound whatever the width)
    static const int one_minc = [] { const char* e = getenv("MVLT_SWIN_BWD_ONE"); const int v = e ? atoi(e) : 384; return v <= 0 ? 1 << 30 : (v == 1 ? 96 : v); }();
    const int nparts = (C >= one_minc && (shift == 0 || shift == 3)) ? mvlt_swin_wmsa2_bwd_parts(dtype, B, res, C, nH) : 0;
    Tensor dao = dyw;
    if (!fuse_proj && !nparts) {
        dao = empty2(rows, C, x);
        Epi e; e.pf = P(w[0]); e.pf_bytes = (int64_t)3 * C * C * esz; dgrad(dyw, w[1], C, dao, e, st);
    }
    Tensor dqkv = empty2(rows, 3 * C, x);
    Tensor dxn1w;
    if (nparts) {
        dxn1w = at::empty({(int64_t)nparts * rows, (int64_t)C}, x.options());
        const int nwg = mvlt_swin_wmsa2_bwd_workgroups(dtype, B, res, C, nH);
        float* ws = g_lnq.take((int64_t)nwg * 507, x);
        MvltSwinWmsa p{};
        p.dtype = dtype; p.B = B; p.res = res; p.C = C; p.nH = nH; p.shift = shift;
        p.dy_win = dp(dyw); p.qkv_win = dp(qkv); p.lse = fp(lse); p.wproj = P(w[1]); p.wqkv = P(w[0]);
        p.bias_table = P<float>(f[4]); p.scale = (float)scale;
        p.dqkv = dp(dqkv); p.dxn_win = dp(dxn1w); p.dbias_ws = ws;
        g_lnq.dbias.push_back(MvltSwinDbiasItem{ws, nwg, nH, P<float>(g[6])});
        hipEvent_t e = next_event();
        ck(mvlt_swin_wmsa2_bwd_ev(&p, ss.main, e), "mvlt_swin_wmsa2_bwd_ev");
        TORCH_CHECK(hipStreamWaitEvent(static_cast<hipStream_t>(ss.side), e, 0) == hipSuccess, "hipStreamWaitEvent");
    } else
    { MvltAttn p{}; p.dtype = dtype; p.mode = MVLT_ATTN_SWIN; p.nseq = B * nW; p.L = 49; p.nH = nH; p.hd = C / nH;
      p.qkv = dp(qkv); p.out = dp(ao); p.lse = fp(lse); p.scale = (float)scale;
      p.bias_table = P<float>(f[4]); p.nW = nW; p.win_res = res; p.shift = shift;
      p.dout = dp(dao); p.dqkv = dp(dqkv); p.dbias_table = P<float>(g[6]);
      if (fuse_proj) { p.dout_weight = P(w[1]); p.prefetch = P(w[0]); p.prefetch_bytes = (int64_t)3 * C * C * esz; }
      attn_bwd_fork(p, ss); }
    for (const Tensor* t : std::initializer_list<const Tensor*>{&dy2, &act, &dh, &xn2, &dyw, &ao, &dqkv, &xn1w}) g_side_keepalive.push_back(*t);
    wgrad_group({{&dy2, &act, g[11], g[12]}, {&dh, &xn2, g[9], g[10]}, {&dyw, &ao, g[4], g[5]}, {&dqkv, &xn1w, g[2], g[3]}},
                ss.side, g_ws_side);
    if (!nparts) {
        dxn1w = empty2(rows, C, x);
        Epi e; if (nb) { e.pf = P(w[6]); e.pf_bytes = w[7]; } dgrad(dqkv, w[0], C, dxn1w, e, st);
    }
    Tensor dx0 = empty2(rows, C, x), dy2n;
    { LnBranch br;
      if (s2_next) { dy2n = empty2(rows, C, x); br.dz = dp(dy2n); br.rowscale = P<float>(s2_next); br.rps = Lt; }
      ln_bwd(dxn1w, n2w, x, fp(stat1), fp(stat1) + rows, (int)rows, C, f[0], g[0], g[1], dp(dx1), dx0, br, st, nullptr, nparts); }
    if (s2_next) return {dx0, dy2n};
    return {dx0};
}

// ----------------------------------------------------------------------------------------------- housekeeping
void lnq_flush(int64_t stream) { g_lnq.flush(P(stream)); }
int64_t lnq_pending() { return (int64_t)(g_lnq.items.size() + g_lnq.dbias.size()); }
void side_release() { g_side_keepalive.clear(); g_ws_side.retired.clear(); g_ws_main.retired.clear(); }
// a HIP graph has recorded the addresses of the current scratch buffers: keep them alive for the life of the process
std::vector<Tensor> g_pinned_scratch;
void scratch_pin() {
    for (Scratch* s : {&g_ws_main, &g_ws_side}) {
        if (!s->buf.defined()) continue;
        bool have = false;
        for (const Tensor& t : g_pinned_scratch) have = have || t.data_ptr() == s->buf.data_ptr();
        if (!have) g_pinned_scratch.push_back(s->buf);          // once per buffer, however many graphs are captured
    }
}

// which: 0 = grouped weight-gradient launches, 1 = forward / dgrad products issued by the layer calls
void timer_begin(int64_t which, int64_t every, int64_t capacity) {
    (which == 0 ? g_timer : g_timer_fam).reset((int)every, (size_t)capacity);
}
// -> [(executed flops, milliseconds, algorithmic bytes)] of the sampled launches; call after a device synchronize
std::vector<std::tuple<double, double, double>> timer_collect(int64_t which) { return (which == 0 ? g_timer : g_timer_fam).collect(); }

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "mvlt_amd native host path: per-layer launch sequences over the C-ABI of include/mvlt_hip.h";
    m.def("bert_layer_fwd", &bert_layer_fwd);
    m.def("bert_layer_bwd", &bert_layer_bwd);
    m.def("swin_block_fwd", &swin_block_fwd);
    m.def("swin_block_bwd", &swin_block_bwd);
    m.def("lnq_flush", &lnq_flush);
    m.def("lnq_pending", &lnq_pending);
    m.def("side_release", &side_release);
    m.def("scratch_pin", &scratch_pin);
    m.def("timer_begin", &timer_begin);
    m.def("timer_collect", &timer_collect);
    // the ABI this extension was COMPILED against (not what the library it happens to bind reports): ops.host()
    // compares it with the Python mirror's constant and with libmvlt_hip.so's mvlt_version()
    m.def("abi_version", []() { return (int)MVLT_ABI_VERSION; });
    m.def("struct_sizes", []() {
        return std::vector<int64_t>{(int64_t)sizeof(MvltGemm), (int64_t)sizeof(MvltLayerNorm), (int64_t)sizeof(MvltLayerNormBwd),
                                    (int64_t)sizeof(MvltLnReduceItem), (int64_t)sizeof(MvltAttn), (int64_t)sizeof(MvltSwinWmsa),
                                    (int64_t)sizeof(MvltEmbed), (int64_t)sizeof(MvltAttnCached), (int64_t)sizeof(MvltZeroItem),
                                    (int64_t)sizeof(MvltRange), (int64_t)sizeof(MvltMlmMask), (int64_t)sizeof(MvltGreedyState),
                                    (int64_t)sizeof(MvltSwinDbiasItem)};
    });
}
