// Device-side pieces shared by the GEMM translation units (gemm.hip, gemm8.hip): the kernel argument block, the
// fused epilogue (order documented in include/mvlt_hip.h, MvltGemm), the XCD-aware tile order and the weight prefetch.
#pragma once
#include "common.h"

namespace {

struct GemmDev {
    int M, N, K;
    const void* A; long lda; const void* B; long ldb;
    void* C; long ldc;
    int epi;
    const float* bias; void* pre; const void* residual; long ldr; const void* aux;
    const float* rowscale; int rps; const int* rowmap;
    uint32_t drop_thresh; float drop_scale; uint64_t seed; uint32_t tag;
    int split_k; int k_per_split; float* ws;
    int a_vec, b_vec, epi_vec;
    float* a_colsum; float* ws_colsum;   // optional: column sums of a k-major A (bias gradient), fused
    const char* pf; long pf_lines;       // optional: 128-byte lines of the next product's weights to pull towards the caches
    const int* m_dev;                    // optional: valid storage rows of A on the device (MvltGemm.m_dev)
    int atomic_out;                      // grouped weight gradients with in-launch split-K: f32 atomicAdd onto zeroed C
    int a_kmajor;                        // A is k-major (weight gradients): m_dev then limits the reduction, not the rows
    int wide;                            // bf16 rows out with 16-byte aligned 8-column chunks (and operands): kernels built with WIDE
    int xcs;                             // column groups of the tile order (tile_coords below); 0 / 1 = plain row-major list
};

// Ragged batches planned on the GPU: the launch is sized for the upper bound, the kernel reads the real count.
// k-contiguous A: M shrinks (tiles beyond it leave at once); k-major A (weight gradients): the reduction shrinks.
template <bool AK>
MVLT_DEV GemmDev effective(const GemmDev& p) {
    GemmDev q = p;
    if (p.m_dev) {
        const int r = __builtin_amdgcn_readfirstlane(*p.m_dev);
        if (AK) q.K = min(q.K, max(r, 0)); else q.M = min(q.M, max(r, 0));
    }
    return q;
}

template <typename T>
MVLT_DEV typename TypeInfo<T>::Vec load_chunk(const T* base, long ld, int outer, int inner,
                                              int outer_lim, int inner_lim, bool vec_ok) {
    using Vec = typename TypeInfo<T>::Vec;
    constexpr int E = TypeInfo<T>::E;
    Vec v = zero_vec<T>();
    if (outer >= outer_lim || inner >= inner_lim) return v;
    const T* p = base + (long)outer * ld + inner;
    if (vec_ok && inner + E <= inner_lim) return *reinterpret_cast<const Vec*>(p);
#pragma unroll
    for (int e = 0; e < E; ++e)
        if (inner + e < inner_lim) v[e] = p[e];
    return v;
}

template <typename T>
MVLT_DEV void epilogue4(const GemmDev& p, int m, int n, f32x4 v) {
    if (m >= p.M || n >= p.N) return;
    const int epi = p.epi;
    const int mo = (epi & MVLT_EPI_ROWMAP) ? p.rowmap[m] : m;
    const int nv = min(4, p.N - n);
    const bool vec = p.epi_vec && nv == 4;
    T* Ct = reinterpret_cast<T*>(p.C);
    float* Cf = reinterpret_cast<float*>(p.C);
    const long co = (long)mo * p.ldc + n;
    if (epi & MVLT_EPI_BIAS) {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (j < nv) v[j] += p.bias[n + j];
    }
    if (epi & MVLT_EPI_GELU) {
        if (epi & MVLT_EPI_SAVE_PRE) {
            T* pre = reinterpret_cast<T*>(p.pre) + co;
            if (vec) store4f(pre, v);
            else for (int j = 0; j < nv; ++j) pre[j] = from_f<T>(v[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = gelu_f(v[j]);
    }
    if (epi & MVLT_EPI_DROPOUT) {
        const uint32_t base = (uint32_t)m * (uint32_t)p.N + (uint32_t)n;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            v[j] = rng_keep(p.seed, p.tag, base + j, p.drop_thresh) ? v[j] * p.drop_scale : 0.0f;
    }
    if (epi & MVLT_EPI_ROWSCALE) {
        const float s = p.rowscale[mo / p.rps];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= s;
    }
    if (epi & MVLT_EPI_MUL_GELU_GRAD) {
        const T* aux = reinterpret_cast<const T*>(p.aux) + co;
        if (vec) { f32x4 a = load4f(aux);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= gelu_grad_f(a[j]);
        } else for (int j = 0; j < nv; ++j) v[j] *= gelu_grad_f(to_f(aux[j]));
    }
    if (epi & MVLT_EPI_RESIDUAL) {
        const T* r = reinterpret_cast<const T*>(p.residual) + (long)mo * p.ldr + n;
        if (vec) { f32x4 a = load4f(r);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += a[j];
        } else for (int j = 0; j < nv; ++j) v[j] += to_f(r[j]);
    }
    if (epi & MVLT_EPI_OUT_F32) {
        if (epi & MVLT_EPI_ACCUM) for (int j = 0; j < nv; ++j) v[j] += Cf[co + j];
        if (vec) store4f(Cf + co, v);
        else for (int j = 0; j < nv; ++j) Cf[co + j] = v[j];
    } else {
        if (epi & MVLT_EPI_ACCUM) for (int j = 0; j < nv; ++j) v[j] += to_f(Ct[co + j]);
        if (vec) store4f(Ct + co, v);
        else for (int j = 0; j < nv; ++j) Ct[co + j] = from_f<T>(v[j]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Epilogue of a whole wave tile (FM x FN fragments of 16 x 16; a lane owns 4 consecutive columns of one row per fragment),
// same arithmetic and order as epilogue4, but with every LOAD hoisted out of the store sequence.
// Why: vmcnt completes IN ORDER, loads and stores alike.  epilogue4 called fragment by fragment loads bias / residual /
// aux behind the previous fragment's stores, inside divergent (bounds-checked) blocks, where hipcc waits vmcnt(0): every
// fragment then waits for the previous fragment's stores to COMPLETE (~0.7 us under load): +12 us for a bias, +21 us
// for bias + GELU + saved pre-activation on the BERT FFN-in product (gpurun_out g8_check, round 3).  Here the bias values
// are loaded once per tile and the per-row / per-fragment operands one row block ahead, all from clamped (always valid)
// addresses in straight-line code; only the stores are predicated.
// Requires p.epi_vec and N % 4 == 0 (the callers fall back to epilogue4 otherwise); ACCUM keeps its in-place load.
template <typename T> struct Raw4;
template <> struct Raw4<bf16_t> { using type = bf16x4; };
template <> struct Raw4<float> { using type = f32x4; };
MVLT_DEV f32x4 raw4_to_f(const bf16x4& v) { f32x4 r; r[0] = (float)v[0]; r[1] = (float)v[1]; r[2] = (float)v[2]; r[3] = (float)v[3]; return r; }
MVLT_DEV f32x4 raw4_to_f(const f32x4& v) { return v; }

template <typename T, int FM, int FN>
MVLT_DEV void tile_epilogue(const GemmDev& p, const int m_base, const int n_base, f32x4 (&acc)[FM][FN]) {
    using R4 = typename Raw4<T>::type;
    const int lane = threadIdx.x & 63;
    const int mr = lane & 15, nq = 4 * (lane >> 4);
    const int epi = p.epi;
    const bool has_bias = (epi & MVLT_EPI_BIAS) != 0, has_res = (epi & MVLT_EPI_RESIDUAL) != 0,
               has_aux = (epi & MVLT_EPI_MUL_GELU_GRAD) != 0, has_map = (epi & MVLT_EPI_ROWMAP) != 0,
               has_scale = (epi & MVLT_EPI_ROWSCALE) != 0;
    f32x4 bias_v[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) bias_v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (has_bias) {
#pragma unroll
        for (int j = 0; j < FN; ++j) bias_v[j] = *reinterpret_cast<const f32x4*>(p.bias + min(n_base + j * 16 + nq, p.N - 4));
#pragma unroll
        for (int j = 0; j < FN; ++j) asm volatile("" : "+v"(bias_v[j]));          // waited for here, once (see above)
    }
    struct RowPre { int mo; float sc; R4 res[FN]; R4 aux[FN]; };
    auto preload = [&](int i, RowPre& r) {
        const int m = min(m_base + i * 16 + mr, p.M - 1);
        r.mo = m; r.sc = 1.0f;
        if (has_map) r.mo = p.rowmap[m];
        if (has_scale) r.sc = p.rowscale[r.mo / p.rps];
        if (has_res) {
#pragma unroll
            for (int j = 0; j < FN; ++j)
                r.res[j] = *reinterpret_cast<const R4*>(reinterpret_cast<const T*>(p.residual) + (long)r.mo * p.ldr + min(n_base + j * 16 + nq, p.N - 4));
        }
        if (has_aux) {
#pragma unroll
            for (int j = 0; j < FN; ++j)
                r.aux[j] = *reinterpret_cast<const R4*>(reinterpret_cast<const T*>(p.aux) + (long)r.mo * p.ldc + min(n_base + j * 16 + nq, p.N - 4));
        }
    };
    auto pin = [&](RowPre& r) {
        if (has_res) {
#pragma unroll
            for (int j = 0; j < FN; ++j) asm volatile("" : "+v"(r.res[j]));
        }
        if (has_aux) {
#pragma unroll
            for (int j = 0; j < FN; ++j) asm volatile("" : "+v"(r.aux[j]));
        }
        if (has_scale) asm volatile("" : "+v"(r.sc));
        if (has_map) asm volatile("" : "+v"(r.mo));
    };
    const bool rowloads = has_res || has_aux || has_map || has_scale;
    RowPre cur, nxt;
    if (rowloads) preload(0, nxt);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m_base + i * 16 + mr;
        if (rowloads) {
            cur = nxt;
            if (i + 1 < FM) preload(i + 1, nxt);          // next row block's loads go out before this one's stores
            pin(cur);
        } else { cur.mo = m; cur.sc = 1.0f; }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = n_base + j * 16 + nq;
            const bool live = m < p.M && n < p.N;
            f32x4 v = acc[i][j] + bias_v[j];
            const long co = (long)cur.mo * p.ldc + n;
            if (epi & MVLT_EPI_GELU) {
                if ((epi & MVLT_EPI_SAVE_PRE) && live) store4f(reinterpret_cast<T*>(p.pre) + co, v);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
            }
            if (epi & MVLT_EPI_DROPOUT) {
                const uint32_t base = (uint32_t)m * (uint32_t)p.N + (uint32_t)n;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = rng_keep(p.seed, p.tag, base + e, p.drop_thresh) ? v[e] * p.drop_scale : 0.0f;
            }
            if (has_scale) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= cur.sc;
            }
            if (has_aux) {
                const f32x4 a = raw4_to_f(cur.aux[j]);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= gelu_grad_f(a[e]);
            }
            if (has_res) v += raw4_to_f(cur.res[j]);
            if (live) {
                if (epi & MVLT_EPI_OUT_F32) {
                    float* o = reinterpret_cast<float*>(p.C) + co;
                    if (epi & MVLT_EPI_ACCUM) v += load4f(o);
                    store4f(o, v);
                } else {
                    T* o = reinterpret_cast<T*>(p.C) + co;
                    if (epi & MVLT_EPI_ACCUM) v += load4f(o);
                    store4f(o, v);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same epilogue with 16 bytes per lane on every row operand (bf16 output, saved pre-activation, residual, gelu' operand):
// the tail of a wide-output product is bound by the ISSUE of its row loads / stores, not by their bytes (8-byte stores:
// 15-22 % of the launch; cdna_hip_programming.md T21).  A lane of a 16 x 16 accumulator fragment owns columns 4g .. 4g + 3
// of one row (g = lane / 16).  For a PAIR of fragments (j, j + 1) v_permlane16_swap trades fragment j of the odd lane groups
// for fragment j + 1 of the even ones; afterwards group g holds EIGHT consecutive columns of one fragment,
//     16 (j + (g & 1)) + 8 (g >> 1) .. + 7      (first register quartet: the lower four, second: the upper four),
// and everything behind it -- bias, GELU, dropout index, row scale, gelu' operand, residual, stores -- runs in that layout.
// Chosen by the host per launch (GemmDev.wide) as a compile-time variant of the kernels: with both forms behind a run-time
// flag the 64 x 128 kernels lost more in their longer epilogue code than the stores won (30.7 vs 27.0 us on 2900 x 3072 x
// 768; 25.4 us with the choice compiled in).  Same arithmetic, same order as tile_epilogue.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
MVLT_DEV void swap16(f32x4& a, f32x4& b) {
    // inline asm: fed vector elements, hipcc (ROCm 7.2) miscompiles __builtin_amdgcn_permlane16_swap (one dword loaded, the result
    // splat over the vector).  s_nop 1 = the two wait states between a VALU write of an operand and the swap's read.
    float a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %4, %5\n\t"
                 "v_permlane16_swap_b32 %6, %7"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3));
    a = f32x4{a0, a1, a2, a3}; b = f32x4{b0, b1, b2, b3};
}
MVLT_DEV void unpack8(const u32x4& w, f32x4& lo, f32x4& hi) {
    const bf16x8 h = __builtin_bit_cast(bf16x8, w);
#pragma unroll
    for (int e = 0; e < 4; ++e) { lo[e] = (float)h[e]; hi[e] = (float)h[4 + e]; }
}
MVLT_DEV u32x4 pack8(const f32x4& lo, const f32x4& hi) {
    bf16x8 h;
#pragma unroll
    for (int e = 0; e < 4; ++e) { h[e] = (bf16_t)lo[e]; h[4 + e] = (bf16_t)hi[e]; }
    return __builtin_bit_cast(u32x4, h);
}

template <typename T, int FM, int FN>
MVLT_DEV void tile_epilogue_wide(const GemmDev& p, const int m_base, const int n_base, f32x4 (&acc)[FM][FN]) {
    static_assert(sizeof(T) == 2 && FN % 2 == 0, "bf16 rows, fragment pairs");
    constexpr int NP = FN / 2;
    const int lane = threadIdx.x & 63;
    const int mr = lane & 15, g = lane >> 4;
    const int cofs = 16 * (g & 1) + 8 * (g >> 1);          // this lane's chunk inside the 32 columns of a fragment pair
    const int epi = p.epi;
    const bool has_bias = (epi & MVLT_EPI_BIAS) != 0, has_res = (epi & MVLT_EPI_RESIDUAL) != 0,
               has_aux = (epi & MVLT_EPI_MUL_GELU_GRAD) != 0, has_map = (epi & MVLT_EPI_ROWMAP) != 0,
               has_scale = (epi & MVLT_EPI_ROWSCALE) != 0;
    f32x4 bias_lo[NP], bias_hi[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) { bias_lo[q] = f32x4{0.f, 0.f, 0.f, 0.f}; bias_hi[q] = bias_lo[q]; }
    if (has_bias) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const float* b = p.bias + min(n_base + 32 * q + cofs, p.N - 8);
            bias_lo[q] = *reinterpret_cast<const f32x4*>(b); bias_hi[q] = *reinterpret_cast<const f32x4*>(b + 4);
        }
#pragma unroll
        for (int q = 0; q < NP; ++q) { asm volatile("" : "+v"(bias_lo[q])); asm volatile("" : "+v"(bias_hi[q])); }
    }
    struct RowPre { int mo; float sc; u32x4 res[NP]; u32x4 aux[NP]; };
    auto preload = [&](int i, RowPre& r) {
        const int m = min(m_base + i * 16 + mr, p.M - 1);
        r.mo = m; r.sc = 1.0f;
        if (has_map) r.mo = p.rowmap[m];
        if (has_scale) r.sc = p.rowscale[r.mo / p.rps];
        if (has_res) {
#pragma unroll
            for (int q = 0; q < NP; ++q)
                r.res[q] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.residual) + (long)r.mo * p.ldr + min(n_base + 32 * q + cofs, p.N - 8));
        }
        if (has_aux) {
#pragma unroll
            for (int q = 0; q < NP; ++q)
                r.aux[q] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.aux) + (long)r.mo * p.ldc + min(n_base + 32 * q + cofs, p.N - 8));
        }
    };
    auto pin = [&](RowPre& r) {
        if (has_res) {
#pragma unroll
            for (int q = 0; q < NP; ++q) asm volatile("" : "+v"(r.res[q]));
        }
        if (has_aux) {
#pragma unroll
            for (int q = 0; q < NP; ++q) asm volatile("" : "+v"(r.aux[q]));
        }
        if (has_scale) asm volatile("" : "+v"(r.sc));
        if (has_map) asm volatile("" : "+v"(r.mo));
    };
    const bool rowloads = has_res || has_aux || has_map || has_scale;
    RowPre cur, nxt;
    if (rowloads) preload(0, nxt);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m_base + i * 16 + mr;
        if (rowloads) {
            cur = nxt;
            if (i + 1 < FM) preload(i + 1, nxt);          // next row block's loads go out before this one's stores
            pin(cur);
        } else { cur.mo = m; cur.sc = 1.0f; }
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int n = n_base + 32 * q + cofs;
            const bool live = m < p.M && n < p.N;           // (N % 8 == 0: a chunk is inside or outside as a whole)
            f32x4 lo = acc[i][2 * q], hi = acc[i][2 * q + 1];
            swap16(lo, hi);
            lo += bias_lo[q]; hi += bias_hi[q];
            const long co = (long)cur.mo * p.ldc + n;
            if (epi & MVLT_EPI_GELU) {
                if ((epi & MVLT_EPI_SAVE_PRE) && live) *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(p.pre) + co) = pack8(lo, hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) { lo[e] = gelu_f(lo[e]); hi[e] = gelu_f(hi[e]); }
            }
            if (epi & MVLT_EPI_DROPOUT) {
                const uint32_t base = (uint32_t)m * (uint32_t)p.N + (uint32_t)n;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lo[e] = rng_keep(p.seed, p.tag, base + e, p.drop_thresh) ? lo[e] * p.drop_scale : 0.0f;
                    hi[e] = rng_keep(p.seed, p.tag, base + 4 + e, p.drop_thresh) ? hi[e] * p.drop_scale : 0.0f;
                }
            }
            if (has_scale) { lo *= cur.sc; hi *= cur.sc; }
            if (has_aux) {
                f32x4 alo, ahi;
                unpack8(cur.aux[q], alo, ahi);
#pragma unroll
                for (int e = 0; e < 4; ++e) { lo[e] *= gelu_grad_f(alo[e]); hi[e] *= gelu_grad_f(ahi[e]); }
            }
            if (has_res) {
                f32x4 rlo, rhi;
                unpack8(cur.res[q], rlo, rhi);
                lo += rlo; hi += rhi;
            }
            if (live) *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(p.C) + co) = pack8(lo, hi);
        }
    }
}

// LDS-DMA as inline asm (16 bytes per lane to wave-uniform lds_dst + lane * 16): hipcc must NOT know that an LDS-DMA is
// in flight -- knowing it, it puts s_waitcnt vmcnt(0) in front of every ds_read_b64_tr_b16 (the intrinsic carries no alias
// information), which serialises the DMA of tile t+1 with the fragment reads of tile t.  Callers count their vmcnt by
// hand.  M0 (the LDS destination) is compiler-reserved: saved and restored inside the statement.
MVLT_DEV void glds16_asm(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own
// L2), so give every XCD one contiguous chunk of the tile list -> neighbouring tiles (same
// A rows / B columns) hit the same L2.  Bijective for any count; speed only.
MVLT_DEV int xcd_remap(int orig, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

// Position t of the tile list -> (row tile, column tile).  With xcs > 1 the list runs through xcs column groups one after the
// other (all row tiles of the first ceil(gx / xcs) column tiles, then the next group ...; the last group takes what is left), so the
// contiguous eighth of the list that xcd_remap hands an XCD is a BLOCK of (8 / xcs)-th of the rows x one column group instead of a
// row band x ALL columns: an XCD's L2 then pulls 1 / xcs of B and xcs / 8 of A over the fabric instead of all of B and 1 / 8 of A.
// (host side: xcs <= gx, so the last group is never empty)
MVLT_DEV void tile_coords(int t, int gx, int gy, int xcs, int& by, int& bx) {
    if (xcs <= 1) { by = t / gx; bx = t - by * gx; return; }
    const int hx = (gx + xcs - 1) / xcs, per = gy * hx;
    const int ng = (gx + hx - 1) / hx;                       // groups that actually exist (<= xcs)
    const int g = min(t / per, ng - 1), r = t - g * per;
    const int w = g == ng - 1 ? gx - g * hx : hx;
    by = r / w; bx = g * hx + (r - by * w);
}

// MvltGemm.prefetch: every thread of the launch reads one dword of a few 128-byte lines of a byte range that a LATER kernel
// will stream (the next nn.Linear's weights, evicted by the optimizer's sweep since their last use); the value is dropped.
MVLT_DEV unsigned prefetch_lines(const GemmDev& p) {
    unsigned acc = 0;
    if (p.pf) {
        const long nthr = (long)gridDim.x * gridDim.y * gridDim.z * blockDim.x;
        const long g = ((long)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
        for (long l = g; l < p.pf_lines; l += nthr) acc ^= *reinterpret_cast<const unsigned*>(p.pf + (l << 7));
    }
    return acc;
}

}  // namespace
