// Fused attention cores for gfx950: Swin (S)W-MSA windows and the MVLBert
// single-stream BERT attention, forward and backward (MvltAttn in
// include/mvlt_hip.h).
//
// One workgroup (4 waves) = one (sequence, head).  Q, K, V (and dO) of that
// head are staged once into LDS ([token][d], padded rows); every product is an
// MFMA over 16x16 tiles and all intermediates (scores, probabilities, dS) live
// in accumulator registers -- nothing of size L x L ever touches LDS or HBM.
//
// Orientation trick: scores are computed TRANSPOSED (first operand = K rows,
// second = Q rows), so a lane owns 4*NT keys of ONE query: the softmax row
// reduction is in-register plus two xor-shuffles, and the probability
// accumulators are directly the second MFMA operand of O^T = V^T P^T (the
// k-slot <-> key permutation is mirrored on the V^T fragment, which is read
// with the gfx950 transposed LDS read).  The backward pass computes both
// orientations (keys-on-rows for dQ, queries-on-rows for dK/dV) instead of
// transposing dS through LDS.
//
// Masks are never materialised: the Swin shift mask (0/-100) and relative
// position index are computed from (window, slot); the MVLBert key mask from
// text ids; the seq2seq mask from (row, col, obj_end).
#include "common.h"
#include <hip/hip_ext.h>
#include "attn_frag.h"
#include <cstdlib>

namespace {
using namespace mvlt_attn;

struct AttnDev {
    int mode, nseq, L, nH, hd, NT;
    const void* qkv; void* out; float* lse; float scale;
    const float* bias_table; int nW, res, shift;
    const int64_t* text_ids; int T; const uint8_t* image_mask; int obj_end;
    uint32_t drop_thresh; float drop_scale; uint64_t seed; uint32_t tag;
    const void* dout; void* dqkv; float* dbias; float* delta_ws;
    const void* dw;                             // Swin backward: output-projection weight when the kernel applies its transpose itself, or null
    const char* pf; long pf_lines;              // Swin backward (scores-once kernel): byte range a LATER kernel streams (MvltAttn.prefetch)
    const int* row_start; const int* seq_len;   // packed rows (MVLBert modes), or null
    int ld;          // LDS row stride (elements)
    int rows_alloc;  // LDS rows per image
};

// first activation row / length of sequence `seq` (dense [nseq*L] rows unless a packed layout is given)
MVLT_DEV long seq_row0(const AttnDev& p, int seq) { return p.row_start ? (long)p.row_start[seq] : (long)seq * p.L; }
MVLT_DEV int seq_length(const AttnDev& p, int seq) { return p.seq_len ? p.seq_len[seq] : p.L; }

// LDS carve-up (all float-aligned): images Q,K,V,(dO) then small arrays
template <typename T> struct Smem {
    T* q; T* k; T* v; T* d; float* kmask; float* lse; float* delta; float* tbl; float* tblg;
    f32x4* lb;       // Swin backward: per-lane bias values, [orientation A/B][KT][256 threads] (SwinLane)
};
__host__ __device__ inline size_t lane_bias_offset(int rows_alloc, int ld, size_t es, bool bwd) {
    const size_t b = (size_t)(bwd ? 4 : 3) * rows_alloc * ld * es + (3 * (size_t)rows_alloc + 352) * sizeof(float);
    return (b + 15) & ~(size_t)15;
}
template <typename T>
MVLT_DEV Smem<T> carve(char* base, const AttnDev& p, bool bwd) {
    Smem<T> s;
    const size_t img = (size_t)p.rows_alloc * p.ld * sizeof(T);
    s.q = reinterpret_cast<T*>(base); s.k = reinterpret_cast<T*>(base + img);
    s.v = reinterpret_cast<T*>(base + 2 * img); s.d = reinterpret_cast<T*>(base + 3 * img);
    float* f = reinterpret_cast<float*>(base + (bwd ? 4 : 3) * img);
    s.kmask = f; s.lse = f + p.rows_alloc; s.delta = f + 2 * p.rows_alloc;
    s.tbl = f + 3 * p.rows_alloc; s.tblg = s.tbl + 176;
    s.lb = reinterpret_cast<f32x4*>(reinterpret_cast<char*>(base) + lane_bias_offset(p.rows_alloc, p.ld, sizeof(T), bwd));
    return s;
}
static size_t smem_bytes(int dtype, int rows_alloc, int ld, bool bwd, bool swin = false) {
    const size_t es = dtype == MVLT_BF16 ? 2 : 4;
    // Swin backward: + two lane-bias tables of KT(=4) x 256 x 16 B
    return lane_bias_offset(rows_alloc, ld, es, bwd) + ((bwd && swin) ? 2 * 4 * 256 * sizeof(f32x4) : 0);
}

// Register-batched staging of one [rows x HD] head slice: all 16-byte global loads of a slice are issued
// back to back (compile-time trip count, no load->ds_write dependency between iterations), the LDS writes
// happen later -- for the window kernels one sequence ahead, so the next window's loads are in flight
// while the current one is multiplied.
template <typename T, int HD, int ROWS>
struct Stager {
    static constexpr int E = TypeInfo<T>::E;
    static constexpr int CPR = HD / E;                  // 16-byte chunks per row
    static constexpr int IT = (ROWS * CPR + 255) / 256;
    using Vec = typename TypeInfo<T>::Vec;
    Vec r[IT];
    // src -> (token 0 of the sequence, first column of this head); stride = elements per token
    MVLT_DEV void load(const T* src, long stride, int row0, int nrows, int L) {
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int idx = threadIdx.x + 256 * i;
            const int rr = idx / CPR, ch = idx % CPR, tok = row0 + rr;
            Vec v = zero_vec<T>();
            if (rr < nrows && tok < L) v = *reinterpret_cast<const Vec*>(src + (long)tok * stride + ch * E);
            r[i] = v;
        }
    }
    MVLT_DEV void store(T* img, int ld, int nrows) const {
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int idx = threadIdx.x + 256 * i;
            const int rr = idx / CPR, ch = idx % CPR;
            if (rr < nrows) *reinterpret_cast<Vec*>(img + rr * ld + ch * E) = r[i];
        }
    }
};

// additive logit term for (query q, key k); also folds key padding
template <bool SWIN>
MVLT_DEV float logit_bias(const AttnDev& p, const float* kmask, const float* tbl, int q, int k,
                          int wy, int wx, int Ls) {
    if (k >= Ls) return NEG_BIG;
    if (SWIN) {
        float b = tbl[rel_index(min(q, 48), k)];
        // wy < 0 flags an interior window: after the cyclic shift only windows of the last window
        // row / column straddle an image border, every other window's mask is all zeros
        if (wy >= 0 && swin_region(min(q, 48), wy, wx, p.res, p.shift) != swin_region(k, wy, wx, p.res, p.shift))
            b += -100.0f;
        return b;
    }
    if (p.mode == MVLT_ATTN_SEQ2SEQ) return (k <= q || k <= p.obj_end) ? 0.0f : -10000.0f;
    return kmask[k];
}

template <bool SWIN>
MVLT_DEV void stage_small(const AttnDev& p, float* kmask, float* tbl, float* tblg, int seq, int h, bool bwd) {
    if (SWIN) {
        for (int i = threadIdx.x; i < 169; i += 256) {
            tbl[i] = p.bias_table[i * p.nH + h];
        }
    } else if (p.mode == MVLT_ATTN_BIDIR) {
        const int n_img = p.obj_end - 1;
        for (int k = threadIdx.x; k < p.L; k += 256) {
            bool ok = true;
            if (k >= 1 && k <= n_img) ok = p.image_mask ? p.image_mask[(long)seq * n_img + k - 1] != 0 : true;
            else if (k > p.obj_end) ok = p.text_ids[(long)seq * p.T + (k - p.obj_end - 1)] > 0;
            kmask[k] = ok ? 0.0f : -10000.0f;
        }
    }
}

// Swin: the (query, key) a lane's accumulator element belongs to is the same in every window, so the
// relative-position bias of its KT*4 elements and, for shifted blocks, which of them straddle an image border
// when the window lies in the last window row / column, are computed ONCE per workgroup (the per-element
// work in the window loop drops to one fma and a bit test).  Orientation: KEYS_ON_ROWS -> element (t, j) is
// key 16t + 4g + j against query `other`; otherwise query 16t + 4g + j against key `other`.
template <int KT, bool KEYS_ON_ROWS>
struct SwinLane {
    float bias[KT][4];
    uint32_t rowbits, colbits;
    MVLT_DEV void init(const float* tbl, int other, int g, int shift) {
        rowbits = colbits = 0;
        const int oc = min(other, 48);
        const int oy = div7(oc), ox = oc - 7 * oy;
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int tok = 16 * t + 4 * g + j;
                const bool valid = tok < 49 && other < 49;
                const int tc = min(tok, 48);
                const int q = KEYS_ON_ROWS ? oc : tc, k = KEYS_ON_ROWS ? tc : oc;
                bias[t][j] = valid ? tbl[rel_index(q, k)] : NEG_BIG;
                const int ty = div7(tc), tx = tc - 7 * ty;
                // last window row: image rows >= res-7, split at res-shift  <=>  token row < 7-shift or not
                if (valid && ((ty < 7 - shift) != (oy < 7 - shift))) rowbits |= 1u << (4 * t + j);
                if (valid && ((tx < 7 - shift) != (ox < 7 - shift))) colbits |= 1u << (4 * t + j);
            }
    }
};

// ------------------------------------------------------------------ forward
template <typename T, int HD, int KT, bool SWIN>
__global__ __launch_bounds__(256, (SWIN ? 4 : 2)) void attn_fwd_kernel(const AttnDev p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    using M = Mma<T>;
    constexpr int KBD = HD / M::KB;                     // k-blocks over the head dim
    constexpr int TPB = Tok<T>::TPB;
    constexpr int KBT = (KT + TPB - 1) / TPB;           // k-blocks over tokens
    constexpr int TD = HD / 16;
    const Smem<T> s = carve<T>(smem_raw, p, false);
    const int h = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c15 = lane & 15;
    const int C = p.nH * HD;
    T* out = reinterpret_cast<T*>(p.out);
    const T* qkv_g = reinterpret_cast<const T*>(p.qkv);
    Stager<T, HD, KT * 16> gq, gk, gv;
    auto issue = [&](int sq) {
        const long r0 = seq_row0(p, sq);
        const int ln = seq_length(p, sq);
        const T* base = qkv_g + r0 * 3 * C + h * HD;
        gq.load(base, 3 * C, 0, p.rows_alloc, ln);
        gk.load(base + C, 3 * C, 0, p.rows_alloc, ln);
        gv.load(base + 2 * C, 3 * C, 0, p.rows_alloc, ln);
    };
    if (SWIN && (int)blockIdx.x < p.nseq) issue(blockIdx.x);
    SwinLane<SWIN ? KT : 1, true> sl;
    if (SWIN) {
        if (p.shift == 0 || p.shift == 3) {
            // NT == 4 == waves: wave w owns query tile w.  The relative-position indices and shift-mask bit sets of a
            // thread's 16 score elements are constants (SWIN_PAIRS); the bias values are gathered from the table column
            // of this head -- no LDS staging, no barrier, none of the ~400 instructions of SwinLane::init
            const u32x4 idx4 = *reinterpret_cast<const u32x4*>(SWIN_PAIRS.ridx[threadIdx.x]);
#pragma unroll
            for (int t = 0; t < (SWIN ? KT : 1); ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int idx = (idx4[t & 3] >> (8 * j)) & 255;
                    const float b = p.bias_table[min(idx, 168) * p.nH + h];
                    sl.bias[t][j] = idx < 169 ? b : NEG_BIG;
                }
            const uint32_t bb = p.shift ? SWIN_PAIRS.bits3[threadIdx.x] : 0u;
            sl.rowbits = bb & 0xffffu; sl.colbits = bb >> 16;
        } else {
            stage_small<SWIN>(p, s.kmask, s.tbl, s.tblg, 0, h, false);     // bias-table column of this head: once
            __syncthreads();
            sl.init(s.tbl, 16 * wave + c15, g, p.shift);
        }
    }
    for (int seq = blockIdx.x; seq < p.nseq; seq += gridDim.x) {
        const long rs = seq_row0(p, seq);
        const int Ls = seq_length(p, seq);
        const int nt = p.seq_len ? (Ls + 15) >> 4 : p.NT;
        if (!SWIN) issue(seq);
        __syncthreads();
        gq.store(s.q, p.ld, p.rows_alloc);
        gk.store(s.k, p.ld, p.rows_alloc);
        gv.store(s.v, p.ld, p.rows_alloc);
        if (!SWIN) stage_small<SWIN>(p, s.kmask, s.tbl, s.tblg, seq, h, false);
        __syncthreads();
        if (SWIN && seq + (int)gridDim.x < p.nseq) issue(seq + gridDim.x);      // next window in flight
        int wy = 0, wx = 0;
        uint32_t mbits = 0;        // Swin: elements that get the -100 shift mask in this window (:318-344)
        if (SWIN) {
            const int w = seq % p.nW, nwx = p.res / 7;
            wy = w / nwx; wx = w % nwx;
            if (p.shift != 0) mbits = (wy == nwx - 1 ? sl.rowbits : 0u) | (wx == nwx - 1 ? sl.colbits : 0u);
        }
        for (int tq = wave; tq < nt; tq += 4) {
            f32x4 acc[KT];
#pragma unroll
            for (int t = 0; t < KT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            typename M::Frag fq[KBD];
#pragma unroll
            for (int kb = 0; kb < KBD; ++kb) fq[kb] = frag_rowmajor<T>(s.q, p.ld, 16 * tq, kb * M::KB);
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                if (t < nt) {
#pragma unroll
                    for (int kb = 0; kb < KBD; ++kb)
                        M::mma(acc[t], frag_rowmajor<T>(s.k, p.ld, 16 * t, kb * M::KB), fq[kb]);
                }
            }
            const int q = 16 * tq + c15;
            float mx = NEG_BIG;
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v;
                    if (SWIN) {
                        v = fmaf(acc[t][j], p.scale, sl.bias[SWIN ? t : 0][j]);
                        if (mbits & (1u << (4 * t + j))) v -= 100.0f;
                    } else {
                        const int k = 16 * t + 4 * g + j;
                        v = (t < nt) ? acc[t][j] * p.scale + logit_bias<SWIN>(p, s.kmask, s.tbl, q, k, wy, wx, Ls) : NEG_BIG;
                    }
                    acc[t][j] = v;
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float e = __expf(acc[t][j] - mx); acc[t][j] = e; sum += e; }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.0f / sum;
            if (g == 0 && q < Ls && p.lse) p.lse[((long)seq * p.nH + h) * p.L + q] = mx + __logf(sum);
            const bool drop = p.drop_thresh != 0;
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float pr = acc[t][j] * inv;
                    if (drop) {
                        const int k = 16 * t + 4 * g + j;
                        const uint32_t idx = (uint32_t)((((long)seq * p.nH + h) * p.L + q) * p.L + k);
                        pr = rng_keep(p.seed, p.tag, idx, p.drop_thresh) ? pr * p.drop_scale : 0.0f;
                    }
                    acc[t][j] = pr;
                }
            f32x4 o[TD];
#pragma unroll
            for (int td = 0; td < TD; ++td) o[td] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < KBT; ++kb) {
                if (kb * TPB < nt) {
                    const typename M::Frag fp = frag_acc<KT>(acc, kb, T());
#pragma unroll
                    for (int td = 0; td < TD; ++td) M::mma(o[td], frag_tok(s.v, p.ld, 16 * td, kb), fp);
                }
            }
            if (q < Ls) {
#pragma unroll
                for (int td = 0; td < TD; ++td)
                    store4f(out + (rs + q) * C + h * HD + 16 * td + 4 * g, o[td]);
            }
        }
    }
}

// ------------------------------------------------------------------ backward
template <typename T, int HD, int KT, bool SWIN>
__global__ __launch_bounds__(256, (SWIN ? 2 : 1)) void attn_bwd_kernel(const AttnDev p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    using M = Mma<T>;
    constexpr int KBD = HD / M::KB;
    constexpr int TPB = Tok<T>::TPB;
    constexpr int KBT = (KT + TPB - 1) / TPB;
    constexpr int TD = HD / 16;
    const Smem<T> s = carve<T>(smem_raw, p, true);
    const int h = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c15 = lane & 15;
    const int C = p.nH * HD;
    T* dqkv = reinterpret_cast<T*>(p.dqkv);
    const T* outp = reinterpret_cast<const T*>(p.out);
    const T* dout = reinterpret_cast<const T*>(p.dout);
    const bool drop = p.drop_thresh != 0;
    if (SWIN) { for (int i = threadIdx.x; i < 176; i += 256) s.tblg[i] = 0.f; }
    // Swin: every window maps (q,k) to the same lane/register, so dBias[relidx(q,k)] += dS[q,k] is summed
    // in registers over all windows this workgroup walks and scattered to the LDS table once at the end.
    f32x4 dbacc[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) dbacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const T* qkv_g = reinterpret_cast<const T*>(p.qkv);
    Stager<T, HD, KT * 16> gq, gk, gv, gd;
    float lse_pre = 0.f;          // lse of the staged sequence, rows threadIdx.x (+256 ...) -- loaded WITH the operands: a load
                                  // issued behind the previous window's dQ/dK/dV stores would wait for them (vmcnt is in order)
    auto issue = [&](int sq) {
        const long r0 = seq_row0(p, sq);
        const int ln = seq_length(p, sq);
        const T* base = qkv_g + r0 * 3 * C + h * HD;
        gq.load(base, 3 * C, 0, p.rows_alloc, ln);
        gk.load(base + C, 3 * C, 0, p.rows_alloc, ln);
        gv.load(base + 2 * C, 3 * C, 0, p.rows_alloc, ln);
        gd.load(dout + r0 * C + h * HD, C, 0, p.rows_alloc, ln);
        if (p.rows_alloc <= 256) lse_pre = (int)threadIdx.x < ln ? p.lse[((long)sq * p.nH + h) * p.L + threadIdx.x] : 0.f;
    };
    if (SWIN && (int)blockIdx.x < p.nseq) issue(blockIdx.x);
    // Swin: window-invariant per-lane bias values of both orientations go to LDS once (registers are full),
    // the shift-mask bit sets stay in 4 registers
    uint32_t rowA = 0, colA = 0, rowB = 0, colB = 0;
    f32x4* lbA = s.lb;
    f32x4* lbB = s.lb + (SWIN ? KT : 0) * 256;
    if (SWIN) {
        stage_small<SWIN>(p, s.kmask, s.tbl, s.tblg, 0, h, true);
        __syncthreads();
        {
            SwinLane<SWIN ? KT : 1, true> la;
            la.init(s.tbl, 16 * wave + c15, g, p.shift);
            rowA = la.rowbits; colA = la.colbits;
#pragma unroll
            for (int t = 0; t < (SWIN ? KT : 1); ++t) lbA[t * 256 + threadIdx.x] = f32x4{la.bias[t][0], la.bias[t][1], la.bias[t][2], la.bias[t][3]};
        }
        {
            SwinLane<SWIN ? KT : 1, false> lbv;
            lbv.init(s.tbl, 16 * wave + c15, g, p.shift);
            rowB = lbv.rowbits; colB = lbv.colbits;
#pragma unroll
            for (int t = 0; t < (SWIN ? KT : 1); ++t) lbB[t * 256 + threadIdx.x] = f32x4{lbv.bias[t][0], lbv.bias[t][1], lbv.bias[t][2], lbv.bias[t][3]};
        }
    }
    for (int seq = blockIdx.x; seq < p.nseq; seq += gridDim.x) {
        const long rs = seq_row0(p, seq);
        const int Ls = seq_length(p, seq);
        const int nt = p.seq_len ? (Ls + 15) >> 4 : p.NT;
        if (!SWIN) issue(seq);
        __syncthreads();
        gq.store(s.q, p.ld, p.rows_alloc);
        gk.store(s.k, p.ld, p.rows_alloc);
        gv.store(s.v, p.ld, p.rows_alloc);
        gd.store(s.d, p.ld, p.rows_alloc);
        if (!SWIN) stage_small<SWIN>(p, s.kmask, s.tbl, s.tblg, seq, h, true);
        // lse_q (delta_q = rowsum(P .* dP) is produced by phase A in registers: no O / dO pre-pass)
        if (p.rows_alloc <= 256) { if ((int)threadIdx.x < p.rows_alloc) s.lse[threadIdx.x] = lse_pre; }
        else for (int q = threadIdx.x; q < p.rows_alloc; q += 256)
            s.lse[q] = q < Ls ? p.lse[((long)seq * p.nH + h) * p.L + q] : 0.f;
        __syncthreads();
        if (SWIN && seq + (int)gridDim.x < p.nseq) issue(seq + gridDim.x);      // next window in flight
        int wy = 0, wx = 0;
        uint32_t mbA = 0, mbB = 0;     // elements under the -100 shift mask in this window, per orientation
        if (SWIN) {
            const int w = seq % p.nW, nwx = p.res / 7;
            wy = w / nwx; wx = w % nwx;
            if (p.shift != 0) {
                mbA = (wy == nwx - 1 ? rowA : 0u) | (wx == nwx - 1 ? colA : 0u);
                mbB = (wy == nwx - 1 ? rowB : 0u) | (wx == nwx - 1 ? colB : 0u);
            }
        }

        // ---- phase A: keys on accumulator rows, one query tile per wave -> dQ, dBias
        for (int tq = wave; tq < nt; tq += 4) {
            f32x4 sc[KT], dp[KT];
#pragma unroll
            for (int t = 0; t < KT; ++t) { sc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[t] = sc[t]; }
            typename M::Frag fq[KBD], fd[KBD];
#pragma unroll
            for (int kb = 0; kb < KBD; ++kb) {
                fq[kb] = frag_rowmajor<T>(s.q, p.ld, 16 * tq, kb * M::KB);
                fd[kb] = frag_rowmajor<T>(s.d, p.ld, 16 * tq, kb * M::KB);
            }
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                if (t < nt) {
#pragma unroll
                    for (int kb = 0; kb < KBD; ++kb) {
                        M::mma(sc[t], frag_rowmajor<T>(s.k, p.ld, 16 * t, kb * M::KB), fq[kb]);
                        M::mma(dp[t], frag_rowmajor<T>(s.v, p.ld, 16 * t, kb * M::KB), fd[kb]);
                    }
                }
            }
            const int q = 16 * tq + c15;
            const float lse_q = s.lse[q];
            // pass 1: probabilities and (dropout-scaled) dP in place; delta_q = sum_k P*dP
            float dl = 0.f;
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = 16 * t + 4 * g + j;
                    float pr = 0.f, dpv = 0.f;
                    if (SWIN) {        // invalid (q, k) carry bias -1e30 -> p = 0; V rows beyond L are zero in LDS
                        float lg = fmaf(sc[t][j], p.scale, lbA[(SWIN ? t : 0) * 256 + threadIdx.x][j]);
                        if (mbA & (1u << (4 * t + j))) lg -= 100.0f;
                        pr = __expf(lg - lse_q);
                        dpv = dp[t][j];
                        if (drop) {
                            const uint32_t idx = (uint32_t)((((long)seq * p.nH + h) * p.L + q) * p.L + k);
                            dpv = rng_keep(p.seed, p.tag, idx, p.drop_thresh) ? dpv * p.drop_scale : 0.0f;
                        }
                        dl += pr * dpv;
                    } else if (t < nt && k < Ls && q < Ls) {
                        const float lg = sc[t][j] * p.scale + logit_bias<SWIN>(p, s.kmask, s.tbl, q, k, wy, wx, Ls);
                        pr = __expf(lg - lse_q);
                        dpv = dp[t][j];
                        if (drop) {
                            const uint32_t idx = (uint32_t)((((long)seq * p.nH + h) * p.L + q) * p.L + k);
                            dpv = rng_keep(p.seed, p.tag, idx, p.drop_thresh) ? dpv * p.drop_scale : 0.0f;
                        }
                        dl += pr * dpv;
                    }
                    sc[t][j] = pr; dp[t][j] = dpv;
                }
            dl += __shfl_xor(dl, 16, 64);
            dl += __shfl_xor(dl, 32, 64);
            if (g == 0) s.delta[q] = dl;
            // pass 2: dS = P (dP - delta); bias gradient accumulates in registers across windows
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float ds = sc[t][j] * (dp[t][j] - dl);
                    if (SWIN) dbacc[t][j] += ds;
                    sc[t][j] = ds * p.scale;
                }
            f32x4 dq[TD];
#pragma unroll
            for (int td = 0; td < TD; ++td) dq[td] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < KBT; ++kb) {
                if (kb * TPB < nt) {
                    const typename M::Frag fs = frag_acc<KT>(sc, kb, T());
#pragma unroll
                    for (int td = 0; td < TD; ++td) M::mma(dq[td], frag_tok(s.k, p.ld, 16 * td, kb), fs);
                }
            }
            if (q < Ls) {
#pragma unroll
                for (int td = 0; td < TD; ++td)
                    store4f(dqkv + (rs + q) * 3 * C + h * HD + 16 * td + 4 * g, dq[td]);
            }
        }

        __syncthreads();      // delta of every query tile is in LDS
        // ---- phase B: queries on accumulator rows, one key tile per wave -> dK, dV
        for (int tk = wave; tk < nt; tk += 4) {
            f32x4 sc[KT], dp[KT];
#pragma unroll
            for (int t = 0; t < KT; ++t) { sc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[t] = sc[t]; }
            typename M::Frag fk[KBD], fv[KBD];
#pragma unroll
            for (int kb = 0; kb < KBD; ++kb) {
                fk[kb] = frag_rowmajor<T>(s.k, p.ld, 16 * tk, kb * M::KB);
                fv[kb] = frag_rowmajor<T>(s.v, p.ld, 16 * tk, kb * M::KB);
            }
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                if (t < nt) {
#pragma unroll
                    for (int kb = 0; kb < KBD; ++kb) {
                        M::mma(sc[t], frag_rowmajor<T>(s.q, p.ld, 16 * t, kb * M::KB), fk[kb]);
                        M::mma(dp[t], frag_rowmajor<T>(s.d, p.ld, 16 * t, kb * M::KB), fv[kb]);
                    }
                }
            }
            const int k = 16 * tk + c15;
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int q = 16 * t + 4 * g + j;
                    float pd = 0.f, ds = 0.f;
                    if (SWIN || (t < nt && k < Ls && q < Ls)) {
                        float lg;
                        if (SWIN) {
                            lg = fmaf(sc[t][j], p.scale, lbB[(SWIN ? t : 0) * 256 + threadIdx.x][j]);
                            if (mbB & (1u << (4 * t + j))) lg -= 100.0f;
                        } else {
                            lg = sc[t][j] * p.scale + logit_bias<SWIN>(p, s.kmask, s.tbl, q, k, wy, wx, Ls);
                        }
                        const float pr = __expf(lg - s.lse[q]);
                        float dpv = dp[t][j];
                        pd = pr;
                        if (drop) {
                            const uint32_t idx = (uint32_t)((((long)seq * p.nH + h) * p.L + q) * p.L + k);
                            const bool keep = rng_keep(p.seed, p.tag, idx, p.drop_thresh);
                            dpv = keep ? dpv * p.drop_scale : 0.0f;
                            pd = keep ? pr * p.drop_scale : 0.0f;
                        }
                        ds = pr * (dpv - s.delta[q]) * p.scale;
                    }
                    sc[t][j] = ds; dp[t][j] = pd;
                }
            f32x4 dk[TD], dv[TD];
#pragma unroll
            for (int td = 0; td < TD; ++td) { dk[td] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[td] = dk[td]; }
#pragma unroll
            for (int kb = 0; kb < KBT; ++kb) {
                if (kb * TPB < nt) {
                    const typename M::Frag fs = frag_acc<KT>(sc, kb, T());
                    const typename M::Frag fp = frag_acc<KT>(dp, kb, T());
#pragma unroll
                    for (int td = 0; td < TD; ++td) {
                        M::mma(dk[td], frag_tok(s.q, p.ld, 16 * td, kb), fs);
                        M::mma(dv[td], frag_tok(s.d, p.ld, 16 * td, kb), fp);
                    }
                }
            }
            if (k < Ls) {
#pragma unroll
                for (int td = 0; td < TD; ++td) {
                    T* base = dqkv + (rs + k) * 3 * C + h * HD + 16 * td + 4 * g;
                    store4f(base + C, dk[td]);
                    store4f(base + 2 * C, dv[td]);
                }
            }
        }
    }
    if (SWIN && p.dbias) {
        // Swin has NT == 4 == number of waves: wave w owns query tile w in every window
        const int tqw = wave;
        if (tqw < p.NT) {
            const int Ls = p.L;
            const int q = 16 * tqw + c15;
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = 16 * t + 4 * g + j;
                    if (q < Ls && k < Ls) atomicAdd(&s.tblg[rel_index(q, k)], dbacc[t][j]);
                }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 169; i += 256) atomicAdd(&p.dbias[i * p.nH + h], s.tblg[i]);
    }
}


// ------------------------------------------------------------------ Swin backward, scores computed once
// attn_bwd_kernel above evaluates the scores, the probabilities and dS twice -- keys on the accumulator rows for dQ,
// queries on the rows for dK / dV -- and that softmax arithmetic (10 vector instructions per element and orientation)
// is what the kernel spends its time on.  Here phase A (one query tile per wave) also leaves P and dS in LDS as bf16
// images [query][key]; phase B (one key tile per wave) reads them back through the transposing LDS read as the
// key-indexed operand of dK^T = Q^T dS and dV^T = dO^T P: 8 MFMAs and 24 LDS reads per wave, no vector arithmetic.
//   * V is only ever a row-major operand: its four fragments go from global memory straight into registers, the LDS
//     holds Q, K, dO, P, dS and the per-lane bias values of ONE orientation (50 KB).
//   * exp(x) = exp2(x log2 e): log2 e is folded into the scale, the bias values, the mask constant and the saved lse.
//   * 1/sqrt(hd) is applied to the dQ / dK accumulators (8 values per lane) instead of the 16 dS values.
constexpr int SW2_LD = 40;                                   // row stride of the [64][32] images (elements)
constexpr int SW2_LDP = 72;                                  // row stride of the [64][64] P / dS images
constexpr int SW2_IMG = 64 * SW2_LD * 2, SW2_PIMG = 64 * SW2_LDP * 2;
constexpr float LOG2E = 1.4426950408889634f;
constexpr size_t SW2_SMEM = 3 * SW2_IMG + 2 * SW2_PIMG + 4 * 256 * sizeof(f32x4) + 64 * 4;
// KS > 0 (round 6): the output projection's dgrad inside this launch.  `dout` is then the gradient of the projection's OUTPUT
// (window order, [nseq * 49, C], C = 32 KS) and `dw` the projection weight [C_out][C_in]: a workgroup keeps the 32 input columns
// of its head as a [32][C + 8] LDS image (read once: the head is fixed for the workgroup's whole walk), and every wave forms ITS
// 16 rows of dO_h = dY W[:, 32 h .. 32 h + 31] at the top of a window -- 2 KS MFMAs whose row operand comes straight from global
// memory into registers one window ahead (KS b128 loads per lane) -- and writes them, rounded to bf16 as the stand-alone
// product would, where the staged dO rows went.  One launch and one [rows, C] round trip less per Swin block.
// The weight image is k-major, [C][SW2_LD] like the Q / K images (16-byte writes, transposing fragment reads); the rows of every
// 32-row block are permuted so that frag_tok's k-slot order (tokens 4 g + e and 16 + 4 g + e) meets the natural order of the row
// operand's 16-byte chunk (k = 8 g + e).
constexpr size_t sw2_smem(int ks) { return SW2_SMEM + (size_t)ks * 32 * SW2_LD * 2; }

#ifdef SW2_TRACE
#define SW2_T(i) do { if (p.delta_ws && threadIdx.x == 0) reinterpret_cast<long long*>(p.delta_ws)[((long)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = wall_clock64(); } while (0)
#else
#define SW2_T(i) do { } while (0)
#endif
template <bool SHIFT, int KS>
__global__ __launch_bounds__(256, KS ? 2 : 3) void swin_attn_bwd2_kernel(const AttnDev p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    using T = bf16_t;
    using M = Mma<T>;
    const int h = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4, c15 = lane & 15;
    const int C = p.nH * 32;
    T* qi = reinterpret_cast<T*>(smem_raw);
    T* ki = reinterpret_cast<T*>(smem_raw + SW2_IMG);
    T* di = reinterpret_cast<T*>(smem_raw + 2 * SW2_IMG);
    T* pi = reinterpret_cast<T*>(smem_raw + 3 * SW2_IMG);
    T* si = reinterpret_cast<T*>(smem_raw + 3 * SW2_IMG + SW2_PIMG);
    f32x4* lbA = reinterpret_cast<f32x4*>(smem_raw + 3 * SW2_IMG + 2 * SW2_PIMG);        // [4 key tiles][256 threads]
    float* lse_s = reinterpret_cast<float*>(lbA + 4 * 256);
    T* wt = reinterpret_cast<T*>(smem_raw + SW2_SMEM);                                   // KS: [32 KS][SW2_LD]
    const T* qkv_g = reinterpret_cast<const T*>(p.qkv);
    const T* dout = reinterpret_cast<const T*>(p.dout);
    T* dqkv = reinterpret_cast<T*>(p.dqkv);
    SW2_T(0);
#ifdef SW2_TRACE
    if (p.delta_ws && threadIdx.x == 0) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        reinterpret_cast<long long*>(p.delta_ws)[((long)blockIdx.y * gridDim.x + blockIdx.x) * 16 + 12] = ((long long)xcc << 32) | hw;
    }
#endif

    // Loads of the next window are in flight while the current one is multiplied.  vmcnt completes in order and the
    // compiler can only count what is issued on every path, so everything here is straight-line: rows are clamped instead
    // of predicated, the window index is clamped instead of the issue being skipped, nothing USES a loaded value before
    // the loop top, and the stores further down are buffer stores whose out-of-range lanes are dropped by the hardware.
    const int srow = min((int)threadIdx.x >> 2, 48), sch = (threadIdx.x & 3) * 8;
    bf16x8 gq, gk, gd;            // this thread's 16-byte chunk of the Q / K / dO rows
    bf16x8 fy[KS ? KS : 1];       // KS: this lane's row of dY (16 wave + c15, clamped), k-slots 32 k + 8 g .. + 7
    bf16x8 fv[4];                 // V fragments: rows = keys 16t + c15, k-slots = d 8g..8g+7 (clamped row: P is 0 beyond key 48)
    float lse_pre = 0.f;
    auto issue = [&](int sq) {
        const long r0 = (long)sq * 49;
        const T* src = qkv_g + (r0 + srow) * 3 * C + h * 32 + sch;
        gq = *reinterpret_cast<const bf16x8*>(src);
        gk = *reinterpret_cast<const bf16x8*>(src + C);
        if constexpr (KS == 0) gd = *reinterpret_cast<const bf16x8*>(dout + (r0 + srow) * C + h * 32 + sch);
        else {
            const T* yr = dout + (r0 + min(16 * wave + c15, 48)) * C + g * 8;
#pragma unroll
            for (int k = 0; k < KS; ++k) fy[k] = *reinterpret_cast<const bf16x8*>(yr + 32 * k);
        }
        lse_pre = p.lse[((long)sq * p.nH + h) * p.L + min((int)threadIdx.x, 48)];
    };
    auto issue_v = [&](int sq) {
        const T* base = qkv_g + (long)sq * 49 * 3 * C + h * 32 + 2 * C + g * 8;
#pragma unroll
        for (int t = 0; t < 4; ++t) fv[t] = *reinterpret_cast<const bf16x8*>(base + (long)min(16 * t + c15, 48) * 3 * C);
    };
    { const int first = min((int)blockIdx.x, p.nseq - 1); issue(first); issue_v(first); }
    // KS: W[o][32 h .. 32 h + 31] requested now, written to LDS behind the bias set-up below (its gathers are in flight meanwhile)
    constexpr int WCH = KS ? (KS * 128 + 255) / 256 : 1;
    bf16x8 wv[WCH];
    if constexpr (KS > 0) {
        const T* wg = reinterpret_cast<const T*>(p.dw) + h * 32;
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const int idx = min((int)threadIdx.x + 256 * i, KS * 128 - 1);
            wv[i] = *reinterpret_cast<const bf16x8*>(wg + (long)(idx >> 2) * C + (idx & 3) * 8);
        }
    }
    const __amdgpu_buffer_rsrc_t dq_rsrc = __builtin_amdgcn_make_buffer_rsrc(dqkv, 0, (int)((long)p.nseq * 49 * 3 * C * 2), 0x00020000);
    auto store_rows = [&](int tok, bool valid, int col, const f32x4& v) {     // 4 bf16 at dqkv[tok][col..col+3]
        bf16x4 r; r[0] = (bf16_t)v[0]; r[1] = (bf16_t)v[1]; r[2] = (bf16_t)v[2]; r[3] = (bf16_t)v[3];
        const uint32_t off = valid ? (uint32_t)(((long)tok * 3 * C + col) * 2) : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, r), dq_rsrc, off, 0, 0);
    };

    // ---- once per workgroup: the bias values of this thread's 16 accumulator elements (times log2 e), gathered from the
    // table column of this head through the constant index table; the shift-mask bit sets (wave w owns query tile w)
    SW2_T(1);
    {
        const u32x4 idx4 = *reinterpret_cast<const u32x4*>(SWIN_PAIRS.ridx[threadIdx.x]);
        float bv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int idx = (idx4[e >> 2] >> (8 * (e & 3))) & 255;
            bv[e] = p.bias_table[min(idx, 168) * p.nH + h];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int idx = (idx4[t] >> (8 * j)) & 255;
                o[j] = idx < 169 ? bv[4 * t + j] * LOG2E : NEG_BIG;
            }
            lbA[t * 256 + threadIdx.x] = o;
        }
    }
    SW2_T(2);
    if constexpr (KS > 0) {
        // (once; the barrier at the top of the first window orders the writes before the reads)
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const int idx = (int)threadIdx.x + 256 * i;
            const int o = idx >> 2, e = o & 7;
            const int r = (o & ~31) + 4 * ((o >> 3) & 3) + (e & 3) + (e >= 4 ? 16 : 0);
            if (idx < KS * 128) *reinterpret_cast<bf16x8*>(wt + r * SW2_LD + (idx & 3) * 8) = wv[i];
        }
    }
    uint32_t rowA = 0, colA = 0;
    if (SHIFT) { const uint32_t b = SWIN_PAIRS.bits3[threadIdx.x]; rowA = b & 0xffffu; colA = b >> 16; }      // shift == 3 (host check)
    f32x4 dbacc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) dbacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float sc2 = p.scale * LOG2E;
    const float MASKL2 = -100.0f * LOG2E;
    const int nwx = p.res / 7;
    SW2_T(3);

    for (int seq = blockIdx.x; seq < p.nseq; seq += gridDim.x) {
        const long rs = (long)seq * 49;
        __syncthreads();                       // phase B of the previous window has read the images
        if (seq == (int)blockIdx.x) SW2_T(4);
        {
            const int row = threadIdx.x >> 2;
            const bool ok = row < 49;
            const bf16x8 z = zero_vec<T>();
            *reinterpret_cast<bf16x8*>(qi + row * SW2_LD + sch) = ok ? gq : z;
            *reinterpret_cast<bf16x8*>(ki + row * SW2_LD + sch) = ok ? gk : z;
            if constexpr (KS == 0) *reinterpret_cast<bf16x8*>(di + row * SW2_LD + sch) = ok ? gd : z;
        }
        if constexpr (KS > 0) {
            // acc[jt][r] <-> (j = 16 jt + 4 g + r, q = c15): dO_h[q][j] = sum_o dY[q][o] W[o][32 h + j]
            f32x4 o2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int k = 0; k < KS; ++k) {
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) M::mma(o2[jt], frag_tok(wt, SW2_LD, 16 * jt, k), fy[k]);
            }
            const int row = 16 * wave + c15;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                bf16x4 r;
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = row < 49 ? (bf16_t)o2[jt][e] : (bf16_t)0.f;
                *reinterpret_cast<bf16x4*>(di + row * SW2_LD + 16 * jt + 4 * g) = r;
            }
        }
        if (threadIdx.x < 64) lse_s[threadIdx.x] = threadIdx.x < 49 ? lse_pre * LOG2E : 0.f;
        __syncthreads();
        if (seq == (int)blockIdx.x) SW2_T(5);
        const int nxt = seq + (int)gridDim.x < p.nseq ? seq + (int)gridDim.x : seq;      // last window: re-read (harmless)
        issue(nxt);                                                        // next window in flight
        uint32_t mb = 0;
        if (SHIFT) { const int w = seq % p.nW; mb = ((w / nwx) == nwx - 1 ? rowA : 0u) | ((w % nwx) == nwx - 1 ? colA : 0u); }

        // ---- phase A: keys on accumulator rows, query tile `wave` on the columns -> dQ, dBias, P and dS images
        {
            const int tq = wave;
            f32x4 sc[4], dp[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { sc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[t] = sc[t]; }
            const bf16x8 fq = frag_rowmajor<T>(qi, SW2_LD, 16 * tq, 0);
            const bf16x8 fd = frag_rowmajor<T>(di, SW2_LD, 16 * tq, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                M::mma(sc[t], frag_rowmajor<T>(ki, SW2_LD, 16 * t, 0), fq);
                M::mma(dp[t], fv[t], fd);
            }
            issue_v(nxt);                                                  // the V registers are free again
            const int q = 16 * tq + c15;
            const float lse_q = lse_s[q];
            float dl = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 bia = lbA[t * 256 + threadIdx.x];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float lg = fmaf(sc[t][j], sc2, bia[j]);
                    if (SHIFT) lg += (mb & (1u << (4 * t + j))) ? MASKL2 : 0.f;
                    const float pr = __builtin_amdgcn_exp2f(lg - lse_q);
                    dl = fmaf(pr, dp[t][j], dl);
                    sc[t][j] = pr;
                }
            }
            dl += __shfl_xor(dl, 16, 64);
            dl += __shfl_xor(dl, 32, 64);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                store4f(pi + q * SW2_LDP + 16 * t + 4 * g, sc[t]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float ds = sc[t][j] * (dp[t][j] - dl);
                    dbacc[t][j] += ds;
                    sc[t][j] = ds;
                }
                store4f(si + q * SW2_LDP + 16 * t + 4 * g, sc[t]);
            }
            f32x4 dq[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const bf16x8 fs = frag_acc<4>(sc, kb, T());
#pragma unroll
                for (int td = 0; td < 2; ++td) M::mma(dq[td], frag_tok(ki, SW2_LD, 16 * td, kb), fs);
            }
#pragma unroll
            for (int td = 0; td < 2; ++td) store_rows((int)rs + q, q < 49, h * 32 + 16 * td + 4 * g, dq[td] * p.scale);
        }
        if (seq == (int)blockIdx.x) SW2_T(6);
        __syncthreads();                       // P and dS of every query tile are in LDS
        if (seq == (int)blockIdx.x) SW2_T(7);

        // ---- phase B: key tile `wave`: dK^T[d, key] = sum_q Q[q, d] dS[q, key], dV^T[d, key] = sum_q dO[q, d] P[q, key]
        {
            const int tk = wave;
            f32x4 dk[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, dv[2] = {dk[0], dk[0]};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const bf16x8 fs = frag_tok(si, SW2_LDP, 16 * tk, kb);
                const bf16x8 fp = frag_tok(pi, SW2_LDP, 16 * tk, kb);
#pragma unroll
                for (int td = 0; td < 2; ++td) {
                    M::mma(dk[td], frag_tok(qi, SW2_LD, 16 * td, kb), fs);
                    M::mma(dv[td], frag_tok(di, SW2_LD, 16 * td, kb), fp);
                }
            }
            const int k = 16 * tk + c15;
#pragma unroll
            for (int td = 0; td < 2; ++td) {
                store_rows((int)rs + k, k < 49, C + h * 32 + 16 * td + 4 * g, dk[td] * p.scale);
                store_rows((int)rs + k, k < 49, 2 * C + h * 32 + 16 * td + 4 * g, dv[td]);
            }
        }
        if (seq == (int)blockIdx.x) SW2_T(8);
    }
    SW2_T(9);
    if (p.pf) {          // MvltAttn.prefetch: one dword of every 128-byte line, dropped
        const long nthr = (long)gridDim.x * gridDim.y * 256;
        unsigned acc = 0;
        for (long l = ((long)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; l < p.pf_lines; l += nthr)
            acc ^= *reinterpret_cast<const unsigned*>(p.pf + (l << 7));
        asm volatile("" :: "v"(acc));
    }
    if (p.dbias) {
        // dBias[rel(q, k)] = sum of dS[q, k] over the windows of this workgroup: the per-lane sums go to LDS as a [q][k]
        // matrix (over the P / dS images), then one thread per table entry adds up its diagonal (LDS float atomics run at
        // about a lane per cycle per CU: 3 us for this with three workgroups on the CU)
        float* mat = reinterpret_cast<float*>(pi);                 // [64][68] f32 = 17,408 B <= 2 * SW2_PIMG
        __syncthreads();
        const int q = 16 * wave + c15;
#pragma unroll
        for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(mat + q * 68 + 16 * t + 4 * g) = dbacc[t];
        __syncthreads();
        SW2_T(10);
        if (threadIdx.x < 169) {
            const int dy = (int)threadIdx.x / 13 - 6, dx = (int)threadIdx.x % 13 - 6;
            // 49 independent, clamped LDS reads (a loop over the valid range only would wait out every read in turn)
            const int shiftk = dy * 7 + dx;
            float sum = 0.f;
#pragma unroll
            for (int qq = 0; qq < 49; ++qq) {
                const int qy = qq / 7, qx = qq % 7;
                const bool ok = qy - dy >= 0 && qy - dy <= 6 && qx - dx >= 0 && qx - dx <= 6;
                const float v = mat[qq * 68 + min(max(qq - shiftk, 0), 48)];
                sum += ok ? v : 0.f;
            }
            atomicAdd(&p.dbias[threadIdx.x * p.nH + h], sum);
        }
    }
    SW2_T(11);
}

// ------------------------------------------------------------------ backward, split in two launches (MVLBert)
// PHASE 0 (dQ + delta): K,V staged in full, Q,dO only the 4 query tiles of this workgroup.
// PHASE 1 (dK, dV)    : Q,dO staged in full, K,V only the 4 key tiles of this workgroup.
// 66 KB of LDS and half the live registers of the fused kernel -> 2 workgroups per CU, 3x the
// workgroups (grid.z = ceil(NT/4)); delta_q = rowsum(P .* dP) goes from phase 0 to phase 1 through
// `delta_ws` in global memory (separate launches: ordinary stream order, no in-kernel hand-off).
static size_t smem_bytes_split(int dtype, int rows_alloc, int ld) {
    const size_t es = dtype == MVLT_BF16 ? 2 : 4;
    return (size_t)(2 * rows_alloc + 2 * 64) * ld * es + 3 * (size_t)rows_alloc * sizeof(float);
}

template <typename T, int HD, int KT, int PHASE>
__global__ __launch_bounds__(256, 2) void attn_bwd_split_kernel(const AttnDev p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    using M = Mma<T>;
    constexpr int KBD = HD / M::KB;
    constexpr int TPB = Tok<T>::TPB;
    constexpr int KBT = (KT + TPB - 1) / TPB;
    constexpr int TD = HD / 16;
    const size_t full = (size_t)p.rows_alloc * p.ld * sizeof(T), part = (size_t)64 * p.ld * sizeof(T);
    T* F1 = reinterpret_cast<T*>(smem_raw);                 // phase 0: K   phase 1: Q
    T* F2 = reinterpret_cast<T*>(smem_raw + full);          // phase 0: V   phase 1: dO
    T* P1 = reinterpret_cast<T*>(smem_raw + 2 * full);      // phase 0: Q   phase 1: K   (4 tiles)
    T* P2 = reinterpret_cast<T*>(smem_raw + 2 * full + part);   // phase 0: dO  phase 1: V
    float* kmask = reinterpret_cast<float*>(smem_raw + 2 * full + 2 * part);
    float* lse_s = kmask + p.rows_alloc;
    float* delta_s = lse_s + p.rows_alloc;
    const int seq = blockIdx.x, h = blockIdx.y, zt0 = blockIdx.z * 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c15 = lane & 15;
    const int C = p.nH * HD;
    T* dqkv = reinterpret_cast<T*>(p.dqkv);
    const bool drop = p.drop_thresh != 0;
    const long rowbase = ((long)seq * p.nH + h) * p.L;
    const long rs = seq_row0(p, seq);
    const int Ls = seq_length(p, seq);
    const int nt = p.seq_len ? (Ls + 15) >> 4 : p.NT;
    {
        // PHASE 0: F1,F2 = K,V   P1,P2 = Q,dO     PHASE 1: F1,F2 = Q,dO   P1,P2 = K,V
        const int C3 = 3 * p.nH * HD, C1 = p.nH * HD;
        const T* qb = reinterpret_cast<const T*>(p.qkv) + rs * C3 + h * HD;
        const T* db = reinterpret_cast<const T*>(p.dout) + rs * C1 + h * HD;
        Stager<T, HD, KT * 16> f1, f2;
        Stager<T, HD, 64> p1, p2;
        if (PHASE == 0) {
            f1.load(qb + C1, C3, 0, p.rows_alloc, Ls);
            f2.load(qb + 2 * C1, C3, 0, p.rows_alloc, Ls);
            p1.load(qb, C3, 16 * zt0, 64, Ls);
            p2.load(db, C1, 16 * zt0, 64, Ls);
        } else {
            f1.load(qb, C3, 0, p.rows_alloc, Ls);
            f2.load(db, C1, 0, p.rows_alloc, Ls);
            p1.load(qb + C1, C3, 16 * zt0, 64, Ls);
            p2.load(qb + 2 * C1, C3, 16 * zt0, 64, Ls);
        }
        f1.store(F1, p.ld, p.rows_alloc);
        f2.store(F2, p.ld, p.rows_alloc);
        p1.store(P1, p.ld, 64);
        p2.store(P2, p.ld, 64);
    }
    stage_small<false>(p, kmask, nullptr, nullptr, seq, h, true);
    for (int q = threadIdx.x; q < p.rows_alloc; q += 256) {
        lse_s[q] = q < Ls ? p.lse[rowbase + q] : 0.f;
        if (PHASE == 1) delta_s[q] = q < Ls ? p.delta_ws[rowbase + q] : 0.f;
    }
    __syncthreads();
    const int tile = zt0 + wave;
    if (tile >= nt) return;
    f32x4 sc[KT], dp[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) { sc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[t] = sc[t]; }
    typename M::Frag f1[KBD], f2[KBD];
#pragma unroll
    for (int kb = 0; kb < KBD; ++kb) {
        f1[kb] = frag_rowmajor<T>(P1, p.ld, 16 * wave, kb * M::KB);
        f2[kb] = frag_rowmajor<T>(P2, p.ld, 16 * wave, kb * M::KB);
    }
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        if (t < nt) {
#pragma unroll
            for (int kb = 0; kb < KBD; ++kb) {
                M::mma(sc[t], frag_rowmajor<T>(F1, p.ld, 16 * t, kb * M::KB), f1[kb]);
                M::mma(dp[t], frag_rowmajor<T>(F2, p.ld, 16 * t, kb * M::KB), f2[kb]);
            }
        }
    }
    if (PHASE == 0) {
        const int q = 16 * tile + c15;
        const float lse_q = lse_s[q];
        float dl = 0.f;
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 16 * t + 4 * g + j;
                float pr = 0.f, dpv = 0.f;
                if (t < nt && k < Ls && q < Ls) {
                    const float lg = sc[t][j] * p.scale + logit_bias<false>(p, kmask, nullptr, q, k, 0, 0, Ls);
                    pr = __expf(lg - lse_q);
                    dpv = dp[t][j];
                    if (drop) dpv = rng_keep(p.seed, p.tag, (uint32_t)((rowbase + q) * p.L + k), p.drop_thresh) ? dpv * p.drop_scale : 0.0f;
                    dl += pr * dpv;
                }
                sc[t][j] = pr; dp[t][j] = dpv;
            }
        dl += __shfl_xor(dl, 16, 64);
        dl += __shfl_xor(dl, 32, 64);
        if (g == 0 && q < Ls) p.delta_ws[rowbase + q] = dl;
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) sc[t][j] = sc[t][j] * (dp[t][j] - dl) * p.scale;
        f32x4 dq[TD];
#pragma unroll
        for (int td = 0; td < TD; ++td) dq[td] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KBT; ++kb) {
            if (kb * TPB < nt) {
                const typename M::Frag fs = frag_acc<KT>(sc, kb, T());
#pragma unroll
                for (int td = 0; td < TD; ++td) M::mma(dq[td], frag_tok(F1, p.ld, 16 * td, kb), fs);
            }
        }
        if (q < Ls) {
#pragma unroll
            for (int td = 0; td < TD; ++td)
                store4f(dqkv + (rs + q) * 3 * C + h * HD + 16 * td + 4 * g, dq[td]);
        }
    } else {
        const int k = 16 * tile + c15;
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = 16 * t + 4 * g + j;
                float pd = 0.f, ds = 0.f;
                if (t < nt && k < Ls && q < Ls) {
                    const float lg = sc[t][j] * p.scale + logit_bias<false>(p, kmask, nullptr, q, k, 0, 0, Ls);
                    const float pr = __expf(lg - lse_s[q]);
                    float dpv = dp[t][j];
                    pd = pr;
                    if (drop) {
                        const bool keep = rng_keep(p.seed, p.tag, (uint32_t)((rowbase + q) * p.L + k), p.drop_thresh);
                        dpv = keep ? dpv * p.drop_scale : 0.0f;
                        pd = keep ? pr * p.drop_scale : 0.0f;
                    }
                    ds = pr * (dpv - delta_s[q]) * p.scale;
                }
                sc[t][j] = ds; dp[t][j] = pd;
            }
        f32x4 dk[TD], dv[TD];
#pragma unroll
        for (int td = 0; td < TD; ++td) { dk[td] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[td] = dk[td]; }
#pragma unroll
        for (int kb = 0; kb < KBT; ++kb) {
            if (kb * TPB < nt) {
                const typename M::Frag fs = frag_acc<KT>(sc, kb, T());
                const typename M::Frag fp = frag_acc<KT>(dp, kb, T());
#pragma unroll
                for (int td = 0; td < TD; ++td) {
                    M::mma(dk[td], frag_tok(F1, p.ld, 16 * td, kb), fs);
                    M::mma(dv[td], frag_tok(F2, p.ld, 16 * td, kb), fp);
                }
            }
        }
        if (k < Ls) {
#pragma unroll
            for (int td = 0; td < TD; ++td) {
                T* base = dqkv + (rs + k) * 3 * C + h * HD + 16 * td + 4 * g;
                store4f(base + C, dk[td]);
                store4f(base + 2 * C, dv[td]);
            }
        }
    }
}

// ------------------------------------------------------------------ MVLBert backward, scores once (bf16, hd 64, L <= 160)
// The two launches above each evaluate scores, probabilities, the dropout decisions and dS -- ~20 vector instructions per
// element, twice.  Here one workgroup of 5 waves owns a (sequence, head); wave w keeps the key tiles w and w + 5 (their K
// and V fragments and the dK / dV accumulators stay in registers) and the workgroup walks the queries in blocks of 32:
//   * scores with the QUERIES on the accumulator rows, so P and dS are directly the second MFMA operand of
//     dV^T += dO^T P and dK^T += Q^T dS (contraction over the block's 32 queries);
//   * delta_q = rowsum(dO o O) from the forward output (equal to rowsum(P o dP), dropout included), computed while the
//     operands are staged: every (query tile, key tile) pair is independent of the others;
//   * dS also goes to LDS as a bf16 image [32 queries][keys]; after a barrier the waves split dQ^T = K^T dS^T by
//     (feature tile, query tile): complete results, no reduction over the waves that own the keys.
// LDS: K image (for the transposing read), double-buffered Q / dO / dS block images, lse, delta, key mask: 60 KB.
// NW waves own 2 NW key tiles: NW = 5 -> up to 160 rows (config #2: L = 131), NW = 6 -> up to 192 rows (config #5: L = 179; round 5).
constexpr int AB2_LD = 72;                                    // row stride (elements) of the [.][64] images
template <int NW> struct Ab2Geom {
    static constexpr int NT = 64 * NW, ROWS = 32 * NW, LDS = ROWS + 8;          // LDS = row stride of the [32][ROWS] dS image
    static constexpr size_t SMEM = (size_t)ROWS * AB2_LD * 2 + 2 * 2 * 32 * AB2_LD * 2 + 2 * 32 * LDS * 2 + 3 * ROWS * 4;
};

template <bool S2S, bool DROP, int NW>
__global__ __launch_bounds__(64 * NW) void bert_attn_bwd2_kernel(const AttnDev p) {
    constexpr int AB2_NW = NW, AB2_NT = Ab2Geom<NW>::NT, AB2_ROWS = Ab2Geom<NW>::ROWS, AB2_LDS = Ab2Geom<NW>::LDS;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    using T = bf16_t;
    using M = Mma<T>;
    T* kimg = reinterpret_cast<T*>(smem_raw);
    T* qblk = kimg + AB2_ROWS * AB2_LD;                        // [2 buffers][32][72]
    T* dblk = qblk + 2 * 32 * AB2_LD;
    T* simg = dblk + 2 * 32 * AB2_LD;                          // [2 buffers][32][168]
    float* lse_s = reinterpret_cast<float*>(simg + 2 * 32 * AB2_LDS);
    float* delta_s = lse_s + AB2_ROWS;
    float* kmask = delta_s + AB2_ROWS;
    const int seq = blockIdx.x, h = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4, c15 = lane & 15;
    const int C = p.nH * 64;
    const long rs = seq_row0(p, seq);
    const int Ls = seq_length(p, seq);
    const int nt = (Ls + 15) >> 4, nqb = (nt + 1) >> 1;
    const long rowbase = ((long)seq * p.nH + h) * p.L;
    const T* qg = reinterpret_cast<const T*>(p.qkv) + rs * 3 * C + h * 64;
    const T* dg = reinterpret_cast<const T*>(p.dout) + rs * C + h * 64;
    const T* og = reinterpret_cast<const T*>(p.out) + rs * C + h * 64;
    T* dqkv = reinterpret_cast<T*>(p.dqkv) + rs * 3 * C + h * 64;
    constexpr float LOG2E_ = 1.4426950408889634f;

    // ---- this wave's key tiles: K and V fragments (rows = keys, k-slots = features) straight from global memory
    bf16x8 kf[2][2], vf[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = min(16 * (wave + AB2_NW * i) + c15, Ls - 1);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            kf[i][kb] = *reinterpret_cast<const bf16x8*>(qg + (long)row * 3 * C + C + kb * 32 + g * 8);
            vf[i][kb] = *reinterpret_cast<const bf16x8*>(qg + (long)row * 3 * C + 2 * C + kb * 32 + g * 8);
        }
    }
    // first query block: this thread's 16-byte chunk of Q and dO (threads 0..255; clamped row)
    const int srow = (threadIdx.x >> 3) & 31, sch = (threadIdx.x & 7) * 8;
    bf16x8 gq, gd;
    auto issue_block = [&](int qb) {
        const long r = min(32 * qb + srow, Ls - 1);
        gq = *reinterpret_cast<const bf16x8*>(qg + r * 3 * C + sch);
        gd = *reinterpret_cast<const bf16x8*>(dg + r * C + sch);
    };
    issue_block(0);

    // ---- K image, delta = rowsum(dO o O), lse, key mask; the dS images start as zeros (tiles nobody owns stay zero)
    for (int i = threadIdx.x; i < 2 * 32 * AB2_LDS / 8; i += AB2_NT) reinterpret_cast<bf16x8*>(simg)[i] = zero_vec<T>();
    for (int u = threadIdx.x; u < AB2_ROWS * 8; u += AB2_NT) {
        const int row = u >> 3, ch = (u & 7) * 8;
        const bool ok = row < Ls;
        const long r = min(row, Ls - 1);
        const bf16x8 kv = *reinterpret_cast<const bf16x8*>(qg + r * 3 * C + C + ch);
        const bf16x8 dv = *reinterpret_cast<const bf16x8*>(dg + r * C + ch);
        const bf16x8 ov = *reinterpret_cast<const bf16x8*>(og + r * C + ch);
        *reinterpret_cast<bf16x8*>(kimg + row * AB2_LD + ch) = ok ? kv : zero_vec<T>();
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) d = fmaf((float)dv[e], (float)ov[e], d);
        d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
        if ((u & 7) == 0) delta_s[row] = ok ? d : 0.f;
    }
    for (int q = threadIdx.x; q < AB2_ROWS; q += AB2_NT) {
        lse_s[q] = q < Ls ? p.lse[rowbase + min(q, p.L - 1)] * LOG2E_ : 1.0e30f;      // a query outside the sequence: P = 0
        bool ok = q < Ls;
        if (!S2S && ok) {
            const int n_img = p.obj_end - 1;
            if (q >= 1 && q <= n_img) ok = p.image_mask ? p.image_mask[(long)seq * n_img + q - 1] != 0 : true;
            else if (q > p.obj_end) ok = p.text_ids[(long)seq * p.T + (q - p.obj_end - 1)] > 0;
        }
        kmask[q] = q < Ls ? (ok ? 0.0f : -10000.0f * LOG2E_) : NEG_BIG;              // a key outside the sequence: P = 0
    }

    f32x4 dk[2][4], dv[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int td = 0; td < 4; ++td) { dk[i][td] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[i][td] = dk[i][td]; }
    const float sc2 = p.scale * LOG2E_;
    const bool own0 = wave < nt, own1 = wave + AB2_NW < nt;

    for (int qb = 0; qb < nqb; ++qb) {
        T* qb_img = qblk + (qb & 1) * 32 * AB2_LD;
        T* db_img = dblk + (qb & 1) * 32 * AB2_LD;
        T* s_img = simg + (qb & 1) * 32 * AB2_LDS;
        if (threadIdx.x < 256) {
            const bool ok = 32 * qb + srow < Ls;
            *reinterpret_cast<bf16x8*>(qb_img + srow * AB2_LD + sch) = ok ? gq : zero_vec<T>();
            *reinterpret_cast<bf16x8*>(db_img + srow * AB2_LD + sch) = ok ? gd : zero_vec<T>();
        }
        __syncthreads();                                   // block staged (first pass: K image, lse, delta, mask too)
        issue_block(min(qb + 1, nqb - 1));                 // next block in flight (last pass: a harmless re-read)

        // ---- scores of the block against this wave's key tiles: queries on accumulator rows
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i == 0 ? own0 : own1) {
                const int kt = wave + AB2_NW * i;
                const int k = 16 * kt + c15;
                f32x4 sc[2], dp[2];
#pragma unroll
                for (int qi = 0; qi < 2; ++qi) {
                    sc[qi] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[qi] = sc[qi];
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        M::mma(sc[qi], frag_rowmajor<T>(qb_img, AB2_LD, 16 * qi, kb * 32), kf[i][kb]);
                        M::mma(dp[qi], frag_rowmajor<T>(db_img, AB2_LD, 16 * qi, kb * 32), vf[i][kb]);
                    }
                }
                const float kb2 = kmask[k];
#pragma unroll
                for (int qi = 0; qi < 2; ++qi) {
                    const int q0 = 32 * qb + 16 * qi + 4 * g;
                    const f32x4 lq = *reinterpret_cast<const f32x4*>(lse_s + q0);
                    const f32x4 dl = *reinterpret_cast<const f32x4*>(delta_s + q0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int q = q0 + j;
                        float bias = kb2;
                        if (S2S) bias += (k <= q || k <= p.obj_end) ? 0.0f : -10000.0f * LOG2E_;
                        const float pr = __builtin_amdgcn_exp2f(fmaf(sc[qi][j], sc2, bias) - lq[j]);
                        float dpv = dp[qi][j], pd = pr;
                        if (DROP) {
                            const bool keep = rng_keep(p.seed, p.tag, (uint32_t)((rowbase + q) * p.L + k), p.drop_thresh);
                            dpv = keep ? dpv * p.drop_scale : 0.0f;
                            pd = keep ? pr * p.drop_scale : 0.0f;
                        }
                        sc[qi][j] = pr * (dpv - dl[j]);
                        dp[qi][j] = pd;
                    }
                    // dS of this (query tile, key tile) -> the block's dS image [query][key]
                    T* dst = s_img + (16 * qi + 4 * g) * AB2_LDS + k;
#pragma unroll
                    for (int j = 0; j < 4; ++j) dst[j * AB2_LDS] = (T)sc[qi][j];
                }
                const bf16x8 fs = frag_acc<2>(sc, 0, T());
                const bf16x8 fp = frag_acc<2>(dp, 0, T());
#pragma unroll
                for (int td = 0; td < 4; ++td) {
                    M::mma(dk[i][td], frag_tok(qb_img, AB2_LD, 16 * td, 0), fs);
                    M::mma(dv[i][td], frag_tok(db_img, AB2_LD, 16 * td, 0), fp);
                }
            }
        }
        __syncthreads();                                   // the block's dS image is complete

        // ---- dQ^T[feature tile, query tile] = sum over key blocks of K^T dS^T: units dealt to the waves
        for (int u = wave; u < 8; u += AB2_NW) {
            const int dt = u & 3, qi = u >> 2;
            f32x4 dq = f32x4{0.f, 0.f, 0.f, 0.f};
            const T* srow_p = s_img + (16 * qi + c15) * AB2_LDS + 4 * g;
            for (int kb = 0; kb < nqb; ++kb) {
                // k-slot (g, e) of frag_tok <-> key 32 kb + 16 (e / 4) + 4 g + e % 4
                const bf16x4 lo = *reinterpret_cast<const bf16x4*>(srow_p + 32 * kb);
                const bf16x4 hi = *reinterpret_cast<const bf16x4*>(srow_p + 32 * kb + 16);
                bf16x8 fb;
                fb[0] = lo[0]; fb[1] = lo[1]; fb[2] = lo[2]; fb[3] = lo[3]; fb[4] = hi[0]; fb[5] = hi[1]; fb[6] = hi[2]; fb[7] = hi[3];
                M::mma(dq, frag_tok(kimg, AB2_LD, 16 * dt, kb), fb);
            }
            const int q = 32 * qb + 16 * qi + c15;
            if (q < Ls) store4f(dqkv + (long)q * 3 * C + 16 * dt + 4 * g, dq * p.scale);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int k = 16 * (wave + AB2_NW * i) + c15;
        if (k < Ls) {
#pragma unroll
            for (int td = 0; td < 4; ++td) {
                T* ob = dqkv + (long)k * 3 * C + 16 * td + 4 * g;
                store4f(ob + C, dk[i][td] * p.scale);
                store4f(ob + 2 * C, dv[i][td]);
            }
        }
    }
}

// mvlt_attn_bwd_ev: the caller's event rides on the LAST kernel of the call as that dispatch's own stop event
// (hipExtLaunchKernelGGL) instead of a marker packet behind it: an event record costs the recording stream ~5 us, the
// bound form ~2.5 us (scripts/event_cost.hip; the order seen by a stream waiting for the event is checked there too).
thread_local hipEvent_t t_stop_event = nullptr;
#define ATTN_LAUNCH_LAST(kernel, grid, block, shmem, stream, ...)                                                       \
    do {                                                                                                                \
        if (t_stop_event) { hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, nullptr, t_stop_event, 0, __VA_ARGS__); t_stop_event = nullptr; } \
        else hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                       \
    } while (0)

static int bert_bwd_form() { return 1; }          // 1 = one-launch backward where it applies (bf16, <= 192 rows); else two launches
template <int NW>
static int launch_bert_bwd2(const AttnDev& d, hipStream_t s) {
    dim3 grid(d.nseq, d.nH);
    const bool s2s = d.mode == MVLT_ATTN_SEQ2SEQ, drop = d.drop_thresh != 0;
    using GM = Ab2Geom<NW>;
    if constexpr (GM::SMEM > 64 * 1024) {
        static const bool attr = [] {
            bool ok = true;
            ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(bert_attn_bwd2_kernel<true, true, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GM::SMEM) == hipSuccess;
            ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(bert_attn_bwd2_kernel<true, false, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GM::SMEM) == hipSuccess;
            ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(bert_attn_bwd2_kernel<false, true, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GM::SMEM) == hipSuccess;
            ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(bert_attn_bwd2_kernel<false, false, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GM::SMEM) == hipSuccess;
            return ok; }();
        if (!attr) return MVLT_ERR_LAUNCH;
    }
#define AB2_LAUNCH(S, D) ATTN_LAUNCH_LAST((bert_attn_bwd2_kernel<S, D, NW>), grid, dim3(GM::NT), GM::SMEM, s, d)
    if (s2s) { if (drop) AB2_LAUNCH(true, true); else AB2_LAUNCH(true, false); }
    else { if (drop) AB2_LAUNCH(false, true); else AB2_LAUNCH(false, false); }
#undef AB2_LAUNCH
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}

template <typename T, int HD, int KT>
int launch_split(const AttnDev& d, int dtype, hipStream_t s) {
    const size_t sh = smem_bytes_split(dtype, d.rows_alloc, d.ld);
    if (sh > 160 * 1024) return MVLT_ERR_UNSUPPORTED;
    dim3 grid(d.nseq, d.nH, ceil_div(d.NT, 4));
    auto k0 = attn_bwd_split_kernel<T, HD, KT, 0>;
    auto k1 = attn_bwd_split_kernel<T, HD, KT, 1>;
    if (sh > 64 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    }
    hipLaunchKernelGGL(k0, grid, dim3(256), sh, s, d);
    ATTN_LAUNCH_LAST(k1, grid, dim3(256), sh, s, d);
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}

template <typename T, int HD, int KT, bool SWIN>
int launch(const AttnDev& d, bool bwd, int dtype, hipStream_t s) {
    const size_t sh = smem_bytes(dtype, d.rows_alloc, d.ld, bwd, SWIN);
    if (sh > 160 * 1024) return MVLT_ERR_UNSUPPORTED;
    int gx = d.nseq;
    if (SWIN) {   // several windows per workgroup: the LDS bias-gradient table is flushed once
        // backward: ~512 workgroups in total, each walking several windows of one head, so the LDS
        // bias-gradient table is flushed with 169 global atomics per workgroup instead of per window
        const int target = (bwd ? 512 : 2048) / (d.nH > 0 ? d.nH : 1);          // (forward: 512 .. 2048 workgroups measured equal)
        if (gx > target) gx = target < 1 ? 1 : target;
    }
    dim3 grid(gx, d.nH);
    if (bwd) {
        auto k = attn_bwd_kernel<T, HD, KT, SWIN>;
        if (sh > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        ATTN_LAUNCH_LAST(k, grid, dim3(256), sh, s, d);
    } else {
        auto k = attn_fwd_kernel<T, HD, KT, SWIN>;
        if (sh > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        hipLaunchKernelGGL(k, grid, dim3(256), sh, s, d);
    }
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}

// Swin backward, bf16, no attention dropout: the scores-once kernel (f32 and dropout keep attn_bwd_kernel)
static int swin_bwd_form() { return 1; }
static int cu_count() {
    static int n = [] { int dev = 0, v = 0; (void)hipGetDevice(&dev);
                        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
                        return v; }();
    return n;
}
static int launch_swin_bwd2(const AttnDev& d, hipStream_t s) {
    // three workgroups per CU are resident; each walks several windows of one head, so the per-lane bias values are
    // set up and the bias-gradient table is flushed once per workgroup
    int gx = d.nseq;
    // measured (B = 32, us at 1 / 2 / 3 workgroups per CU): stage 0 45.8 / 43.1 / 45.3, stage 1 28.9 / 32.8 / 35.2,
    // stage 2 20.9 / 25.2 / 29.8, stage 3 13.8 / 16.2 / 21.4 -- the set-up of a workgroup costs more than a second
    // resident workgroup hides, except where every workgroup walks dozens of windows
    const int per_cu = (long)d.nseq * d.nH > 16L * cu_count() ? 2 : 1;
    const int cap = per_cu * cu_count() / d.nH;
    if (gx > cap) gx = cap < 1 ? 1 : cap;
    dim3 grid(gx, d.nH);
#define SW2_GO(KS_) do {                                                                                                      \
        auto k0 = swin_attn_bwd2_kernel<false, KS_>;                                                                          \
        auto k1 = swin_attn_bwd2_kernel<true, KS_>;                                                                           \
        constexpr size_t sh = sw2_smem(KS_);                                                                                  \
        if (sh > 64 * 1024) {                                                                                                 \
            static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) == hipSuccess && \
                                   hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) == hipSuccess; \
            if (!ok) return MVLT_ERR_LAUNCH;                                                                                  \
        }                                                                                                                     \
        if (d.shift != 0) ATTN_LAUNCH_LAST(k1, grid, dim3(256), sh, s, d);                                                    \
        else ATTN_LAUNCH_LAST(k0, grid, dim3(256), sh, s, d);                                                                 \
    } while (0)
    if (!d.dw) SW2_GO(0);
    else if (d.nH == 3) SW2_GO(3);
    else if (d.nH == 6) SW2_GO(6);
    else if (d.nH == 12) SW2_GO(12);
    else return MVLT_ERR_UNSUPPORTED;
#undef SW2_GO
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}

template <typename T>
int dispatch(AttnDev d, bool bwd, int dtype, hipStream_t s) {
    constexpr int TPB = Tok<T>::TPB;
    d.NT = ceil_div(d.L, 16);
    d.rows_alloc = ceil_div(d.NT, TPB) * TPB * 16;
    d.ld = d.hd + (sizeof(T) == 2 ? 8 : 4);
    if (d.mode == MVLT_ATTN_SWIN) {
        if (d.hd != 32 || d.L != 49) return MVLT_ERR_UNSUPPORTED;
        if (bwd && sizeof(T) == 2 && d.drop_thresh == 0 && swin_bwd_form() && (d.shift == 0 || d.shift == 3) &&
            (double)d.nseq * 49 * 3 * d.nH * 32 * 2 < 2147483648.0) return launch_swin_bwd2(d, s);     // buffer-store range check: < 2 GB
        if (d.dw) return MVLT_ERR_UNSUPPORTED;          // the projection dgrad only rides on the scores-once kernel
        return launch<T, 32, 4, true>(d, bwd, dtype, s);
    }
    if (d.hd != 64) return MVLT_ERR_UNSUPPORTED;
    if (bwd && sizeof(T) == 2 && d.NT <= 10 && bert_bwd_form()) return launch_bert_bwd2<5>(d, s);
    if (bwd && sizeof(T) == 2 && d.NT <= 12 && bert_bwd_form()) return launch_bert_bwd2<6>(d, s);          // up to 192 rows (config #5)
    if (bwd && d.delta_ws) {           // two-launch backward (dQ+delta, then dK/dV): 2 workgroups per CU
        if (d.NT <= 5) return launch_split<T, 64, 5>(d, dtype, s);
        if (d.NT <= 9) return launch_split<T, 64, 9>(d, dtype, s);
        if (d.NT <= 13) return launch_split<T, 64, 13>(d, dtype, s);      // seq 128 (L = 179, BASELINE config #5)
        return MVLT_ERR_UNSUPPORTED;
    }
    if (d.NT <= 5) return launch<T, 64, 5, false>(d, bwd, dtype, s);
    if (d.NT <= 9) return launch<T, 64, 9, false>(d, bwd, dtype, s);
    if (d.NT <= 13 && !bwd) return launch<T, 64, 13, false>(d, bwd, dtype, s);
    return MVLT_ERR_UNSUPPORTED;
}

int run(const MvltAttn* p, bool bwd, void* stream) {
    MVLT_CHECK(p && p->qkv && p->out, MVLT_ERR_ARG);
    MVLT_CHECK(p->nseq > 0 && p->L > 0 && p->nH > 0, MVLT_ERR_ARG);
    MVLT_CHECK(aligned16(p->qkv) && aligned16(p->out), MVLT_ERR_ARG);
    if (p->mode == MVLT_ATTN_SWIN) MVLT_CHECK(p->bias_table && p->nW > 0 && p->win_res % 7 == 0 &&
                                              p->nW == (p->win_res / 7) * (p->win_res / 7), MVLT_ERR_ARG);
    else if (p->mode == MVLT_ATTN_BIDIR) MVLT_CHECK((p->T == 0 || p->text_ids) && p->obj_end + 1 + p->T == p->L, MVLT_ERR_ARG);
    else if (p->mode == MVLT_ATTN_SEQ2SEQ) MVLT_CHECK(p->obj_end < p->L, MVLT_ERR_ARG);
    else return MVLT_ERR_ARG;
    MVLT_CHECK(p->dropout_p >= 0.f && p->dropout_p < 1.f, MVLT_ERR_ARG);
    MVLT_CHECK((double)p->nseq * p->nH * p->L * p->L < 4294967296.0 || p->dropout_p == 0.f, MVLT_ERR_ARG);
    if (bwd) MVLT_CHECK(p->dout && p->dqkv && p->lse && aligned16(p->dout) && aligned16(p->dqkv), MVLT_ERR_ARG);
    AttnDev d{};
    d.mode = p->mode; d.nseq = p->nseq; d.L = p->L; d.nH = p->nH; d.hd = p->hd;
    d.qkv = p->qkv; d.out = p->out; d.lse = p->lse; d.scale = p->scale;
    d.bias_table = p->bias_table; d.nW = p->nW; d.res = p->win_res; d.shift = p->shift;
    d.text_ids = p->text_ids; d.T = p->T; d.image_mask = p->image_mask; d.obj_end = p->obj_end;
    double th = (double)p->dropout_p * 4294967296.0;
    d.drop_thresh = (uint32_t)th;
    d.drop_scale = 1.0f / (1.0f - p->dropout_p);
    d.seed = p->seed; d.tag = p->tag;
    d.dout = p->dout; d.dqkv = p->dqkv; d.dbias = p->dbias_table; d.delta_ws = p->delta_ws;
    if (bwd && p->dout_weight) {
        MVLT_CHECK(p->mode == MVLT_ATTN_SWIN && aligned16(p->dout_weight), MVLT_ERR_ARG);
        if (!(p->dtype == MVLT_BF16 && p->hd == 32 && (p->nH == 3 || p->nH == 6 || p->nH == 12) && p->dropout_p == 0.f)) return MVLT_ERR_UNSUPPORTED;
        d.dw = p->dout_weight;
    }
    if (bwd && p->prefetch && p->prefetch_bytes >= 128) { d.pf = reinterpret_cast<const char*>(p->prefetch); d.pf_lines = (long)(p->prefetch_bytes >> 7); }
    MVLT_CHECK((p->row_start == nullptr) == (p->seq_len == nullptr), MVLT_ERR_ARG);
    MVLT_CHECK(p->row_start == nullptr || p->mode != MVLT_ATTN_SWIN, MVLT_ERR_ARG);
    d.row_start = p->row_start; d.seq_len = p->seq_len;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (p->dtype == MVLT_F32) return dispatch<float>(d, bwd, MVLT_F32, s);
    if (p->dtype == MVLT_BF16) return dispatch<bf16_t>(d, bwd, MVLT_BF16, s);
    return MVLT_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int mvlt_attn_fwd(const MvltAttn* p, void* stream) { return run(p, false, stream); }
extern "C" int mvlt_attn_bwd(const MvltAttn* p, void* stream) { return run(p, true, stream); }
extern "C" int mvlt_attn_bwd_ev(const MvltAttn* p, void* stream, void* event) {
    MVLT_CHECK(event, MVLT_ERR_ARG);
    t_stop_event = reinterpret_cast<hipEvent_t>(event);
    const int rc = run(p, true, stream);
    if (t_stop_event) {                              // no kernel took it (an error return): nothing is recorded
        t_stop_event = nullptr;
        if (rc == MVLT_OK) return hipEventRecord(reinterpret_cast<hipEvent_t>(event), reinterpret_cast<hipStream_t>(stream)) == hipSuccess ? MVLT_OK : MVLT_ERR_LAUNCH;
    }
    return rc;
}
