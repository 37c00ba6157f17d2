// Fused Swin (S)W-MSA for gfx950 (MvltSwinWmsa in include/mvlt_hip.h): the whole attention half of a
// SwinTransformerBlock -- norm1, cyclic shift + window partition, qkv projection, per-head window attention with
// relative-position bias and shift mask, output projection, window reverse + un-shift, DropPath, residual
// (visual_feature_extractor.py:224-254 and :356-384) -- in ONE launch.  The [tokens, 3C] qkv tensor and the
// attention output never make an HBM round trip (training stores them once, write-only, for the backward pass).
//
// One workgroup (NW = 4 or 8 waves) owns one 7x7 window = 49 token rows, padded to 64 = four 16-row MFMA tiles:
//   * the window's rows are gathered from token order (row map), LayerNorm-ed with f32 statistics and kept in LDS
//     as the A operand of every projection;
//   * heads are walked in groups of G: the group's 96*G qkv columns are projected with the N tiles dealt to the
//     waves, so every weight element is loaded exactly once per workgroup, straight from L2 into MFMA B fragments
//     (no LDS staging, no barrier inside the k-loop: the A tile is static).  The weight fragments run PD k-steps
//     ahead of the MFMAs in a register ring, and the ring is primed across phases: the first k-steps of the next
//     head group are requested before the output-projection slice of the current one, the output-projection
//     fragments before the attention phase -- a workgroup streams 8 C^2 bytes of weights from L2 and is
//     otherwise bound by that latency;
//   * q, k, v of the group go to an LDS tile; each wave then runs (head, query tile) units with the same
//     in-register transposed-score softmax as attn.hip (bias values from an LDS copy of the table through
//     window-invariant per-lane indices);
//   * the heads' output tile O[64, 32G] is multiplied into the output projection immediately
//     (y += O_g Wproj[:, g]^T): the projection accumulators live in registers across the head loop;
//   * epilogue: + bias, DropPath scale, + shortcut, scattered back to token order.
// Two barriers per head group.
#include "common.h"
#include "attn_frag.h"
#include <stdlib.h>

namespace {
using namespace mvlt_attn;

struct WmsaDev {
    int nwin, nW, res, shift, nH;
    const void* x; void* y; const int* w2n;
    const float* gamma; const float* beta; float eps;
    const void* wqkv; const float* bqkv; const void* wproj; const float* bproj;
    const float* bias_table; float scale;
    const float* rowscale;
    void* xn; void* ao; void* qkv; float* lse; float* mean; float* rstd;
    int head_split;          // 1: one workgroup per (window, head group), no output projection (attn_out is the result)
};

template <typename T, int C, int G> struct WmsaGeom {
    static constexpr int PAD = sizeof(T) == 2 ? 8 : 4;
    static constexpr int LDX = C + PAD, LDH = 32 + PAD, LDO = G * 32 + PAD;
    static constexpr size_t XB = (size_t)64 * LDX * sizeof(T);
    static constexpr size_t QB = (size_t)3 * G * 64 * LDH * sizeof(T);
    static constexpr size_t OB = (size_t)64 * LDO * sizeof(T);
    static constexpr int TBL = 176;                               // floats per head in the LDS bias table
    static constexpr size_t bytes(int nH) { return XB + QB + OB + (size_t)nH * TBL * sizeof(float); }
};

// window-invariant, head-invariant facts about the 16 (key, query) pairs a lane's score registers hold
// (keys on accumulator rows: element (t, j) = key 16t + 4g + j against query `query`)
struct LanePairs {
    uint32_t ridx[4];             // 4 x 8-bit relative position indices per key tile (169 = invalid -> -1e30)
    uint32_t rowbits, colbits;    // pairs that straddle the image border in the last window row / column
    MVLT_DEV void init(int query, int g, int shift) {
        rowbits = colbits = 0;
        const int oc = min(query, 48);
        const int oy = div7(oc), ox = oc - 7 * oy;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            ridx[t] = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int key = 16 * t + 4 * g + j;
                const bool valid = key < 49 && query < 49;
                const int kc = min(key, 48);
                const int ky = div7(kc), kx = kc - 7 * ky;
                ridx[t] |= (uint32_t)(valid ? rel_index(oc, kc) : 169) << (8 * j);
                if (valid && ((ky < 7 - shift) != (oy < 7 - shift))) rowbits |= 1u << (4 * t + j);
                if (valid && ((kx < 7 - shift) != (ox < 7 - shift))) colbits |= 1u << (4 * t + j);
            }
        }
    }
};

template <typename T, int C, int G, int NW, int MINW = 1>
__global__ __launch_bounds__(64 * NW, MINW) void wmsa_fwd_kernel(const WmsaDev p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    using GM = WmsaGeom<T, C, G>;
    using M = Mma<T>;
    using Frag = typename M::Frag;
    using Vec = typename TypeInfo<T>::Vec;
    constexpr int NT = 64 * NW;
    constexpr int E = TypeInfo<T>::E, KB = M::KB;
    constexpr int LDX = GM::LDX, LDH = GM::LDH, LDO = GM::LDO;
    constexpr int KSTEPS = C / KB;               // k-steps of the qkv projection
    constexpr int KBD = 32 / KB;                 // k-steps over the head dim
    constexpr int TPB = Tok<T>::TPB;
    constexpr int KBT = 4 / TPB;                 // k-steps over the 64 (padded) keys
    constexpr int NTQ = 6 * G, NTQW = (NTQ + NW - 1) / NW;     // qkv N tiles of a head group, per wave
    constexpr int NTP = C / 16, NTPW = (NTP + NW - 1) / NW;     // proj N tiles, per wave
    constexpr int KSO = G * 32 / KB;             // k-steps of a head group's slice of the output projection
    constexpr int NHG = C / 32 / G;
    constexpr int PD = KSTEPS < 4 ? KSTEPS : 4;  // k-steps the qkv weight fragments run ahead
    constexpr int UPW = 4 * G / NW;              // (head, query tile) attention units per wave
    static_assert(C % 32 == 0 && (C / 32) % G == 0, "head groups");
    static_assert((4 * G) % NW == 0, "attention units per wave");

    T* xln = reinterpret_cast<T*>(smem_raw);
    T* qkvt = reinterpret_cast<T*>(smem_raw + GM::XB);
    T* ot = reinterpret_cast<T*>(smem_raw + GM::XB + GM::QB);
    float* tbl = reinterpret_cast<float*>(smem_raw + GM::XB + GM::QB + GM::OB);

    // head_split: stage-2-sized launches (128 windows at B=32) cannot fill 256 CUs with one workgroup per window and
    // every workgroup would stream all 8 C^2 bytes of weights; there the head groups of a window go to separate
    // workgroups (each streams its own 6 C x 32 G slice) and the output projection is left to a GEMM launch
    const int win = p.head_split ? blockIdx.x / NHG : blockIdx.x;
    const int hg0 = p.head_split ? blockIdx.x % NHG : 0, hg_end = p.head_split ? hg0 + 1 : NHG;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c15 = lane & 15;
    const T* xg = reinterpret_cast<const T*>(p.x);
    const T* wq = reinterpret_cast<const T*>(p.wqkv);
    const T* wp = reinterpret_cast<const T*>(p.wproj);
    const int nH = C / 32;

    // k-slot permutation (bf16, C % 64 == 0): the contraction does not care which 8 inputs a lane's k-slots hold as
    // long as both operands agree, so lane group g takes bytes [32 g, 32 g + 32) of every 128-byte line of a weight
    // row -- first half at the even k-step, second half at the odd one.  Two back-to-back loads then consume whole
    // cache lines (a plain [8 g, 8 g + 8) layout touches every line twice, PD k-steps apart, and the 32-KB L1 has
    // long dropped it by then: twice the L2 traffic).  koff(kk) = element offset of a lane's 16 bytes at step kk.
    constexpr bool PAIR = sizeof(T) == 2 && C % 64 == 0;
    auto koff = [&](int kk) -> int { return PAIR ? (kk >> 1) * 64 + g * 16 + (kk & 1) * 8 : kk * KB + g * E; };
    // weight fragment of qkv tile t of head group hg at k-step kk: 16 output columns x KB inputs, 16 B per lane
    auto wq_ptr = [&](int hg, int t) -> const T* {
        const int tt = min(t, NTQ - 1);
        const int part = tt / (2 * G), within = tt - part * 2 * G;
        return wq + (long)(part * C + hg * G * 32 + within * 16 + c15) * C;
    };
    constexpr bool PPAIR = sizeof(T) == 2 && (G * 32) % 64 == 0;       // same for the projection's k-slices
    auto poff = [&](int ks) -> int { return PPAIR ? (ks >> 1) * 64 + g * 16 + (ks & 1) * 8 : ks * KB + g * E; };
    Frag fb[PD][NTQW];
    // ---- the first weight fragments of head group 0 are requested before anything else
    {
#pragma unroll
        for (int jj = 0; jj < NTQW; ++jj) {
            const T* w = wq_ptr(hg0, wave + NW * jj);
#pragma unroll
            for (int d = 0; d < PD; ++d) fb[d][jj] = *reinterpret_cast<const Frag*>(w + koff(d));
        }
    }

    // ---- bias table -> LDS, [head][176] (entries 169.. = -1e30: padded keys / queries)
    for (int i = threadIdx.x; i < nH * GM::TBL; i += NT) {
        const int h = i / GM::TBL, e = i - h * GM::TBL;
        tbl[i] = e < 169 ? p.bias_table[e * nH + h] : NEG_BIG;
    }

    // ---- gather the window's rows (token order -> window order), LayerNorm, normalised tile -> LDS
    {
        constexpr int LPR = NW;                            // lanes per row (64 rows)
        constexpr int CPR = C / E, CPL = CPR / LPR;        // 16-byte chunks per row / per lane
        static_assert(CPR % LPR == 0, "row chunks");
        const int row = threadIdx.x / LPR, sub = threadIdx.x % LPR;
        const bool rv = row < 49;
        const int tok = rv ? p.w2n[win * 49 + row] : 0;
        const T* src = xg + (long)tok * C;
        Vec xv[CPL];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            xv[i] = rv ? *reinterpret_cast<const Vec*>(src + (sub + LPR * i) * E) : zero_vec<T>();
#pragma unroll
            for (int e = 0; e < E; ++e) s += to_f(xv[i][e]);
        }
#pragma unroll
        for (int o = 1; o < LPR; o <<= 1) s += __shfl_xor(s, o, 64);
        const float mean = s / C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < CPL; ++i)
#pragma unroll
            for (int e = 0; e < E; ++e) { const float d = to_f(xv[i][e]) - mean; q += d * d; }
#pragma unroll
        for (int o = 1; o < LPR; o <<= 1) q += __shfl_xor(q, o, 64);
        const float rstd = rsqrtf(q / C + p.eps);
        if (rv && sub == 0 && p.mean && hg0 == 0) { p.mean[tok] = mean; p.rstd[tok] = rstd; }
        T* xs = (p.xn && hg0 == 0) ? reinterpret_cast<T*>(p.xn) + ((long)win * 49 + row) * C : nullptr;
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int c = (sub + LPR * i) * E;
            Vec o;
#pragma unroll
            for (int e4 = 0; e4 < E; e4 += 4) {
                const f32x4 ga = load4f(p.gamma + c + e4), be = load4f(p.beta + c + e4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[e4 + e] = rv ? from_f<T>((to_f(xv[i][e4 + e]) - mean) * rstd * ga[e] + be[e]) : from_f<T>(0.f);
            }
            *reinterpret_cast<Vec*>(xln + row * LDX + c) = o;
            if (xs && rv) *reinterpret_cast<Vec*>(xs + c) = o;
        }
    }

    // ---- per-lane pair facts for this wave's query tile; shift-mask bits of this window
    const int qt = wave & 3;                      // this wave's query tile in every attention unit it runs
    LanePairs lp;
    if (p.shift == 0 || p.shift == 3) {           // the constants of attn_frag.h instead of ~400 instructions per thread and window
        const int e = qt * 64 + lane;
#pragma unroll
        for (int t = 0; t < 4; ++t) lp.ridx[t] = SWIN_PAIRS.ridx[e][t];
        const uint32_t bb = p.shift ? SWIN_PAIRS.bits3[e] : 0u;
        lp.rowbits = bb & 0xffffu; lp.colbits = bb >> 16;
    } else {
        lp.init(16 * qt + c15, g, p.shift);
    }
    uint32_t mbits = 0;
    if (p.shift != 0) {
        const int w = win % p.nW, nwx = p.res / 7;
        const int wy = w / nwx, wx = w - wy * nwx;
        mbits = (wy == nwx - 1 ? lp.rowbits : 0u) | (wx == nwx - 1 ? lp.colbits : 0u);
    }
    const int qrow = 16 * qt + c15;

    f32x4 pacc[4][NTPW];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < NTPW; ++jj) pacc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};

    __syncthreads();

#pragma unroll 1
    for (int hg = hg0; hg < hg_end; ++hg) {
        // ================= qkv projection of head group hg: [64, C] x [C, 96 G]
        {
            f32x4 acc[4][NTQW];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < NTQW; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
            const T* wrow[NTQW];
#pragma unroll
            for (int jj = 0; jj < NTQW; ++jj) wrow[jj] = wq_ptr(hg, wave + NW * jj);
            const T* arow = xln + c15 * LDX;
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                Frag fa[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const Frag*>(arow + 16 * i * LDX + koff(kk));
#pragma unroll
                for (int jj = 0; jj < NTQW; ++jj) {
                    if (wave + NW * jj < NTQ) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) M::mma(acc[i][jj], fb[kk % PD][jj], fa[i]);
                    }
                    if (kk + PD < KSTEPS) fb[kk % PD][jj] = *reinterpret_cast<const Frag*>(wrow[jj] + koff(kk + PD));
                }
            }
            // + bias, to the q/k/v LDS tiles [part][head][row][32]
#pragma unroll
            for (int jj = 0; jj < NTQW; ++jj) {
                const int t = wave + NW * jj;
                if (t < NTQ) {
                    const int part = t / (2 * G), within = t - part * 2 * G;
                    const f32x4 b4 = load4f(p.bqkv + part * C + hg * G * 32 + within * 16 + 4 * g);
                    T* dst = qkvt + ((part * G + (within >> 1)) * 64 + c15) * LDH + (within & 1) * 16 + 4 * g;
#pragma unroll
                    for (int i = 0; i < 4; ++i) store4f(dst + 16 * i * LDH, acc[i][jj] + b4);
                }
            }
        }
        // output-projection weight fragments of this head group: requested now, used after the attention phase
        Frag fpj[KSO][NTPW];
        if (!p.head_split)
#pragma unroll
        for (int jj = 0; jj < NTPW; ++jj) {
            const int t = min(wave + NW * jj, NTP - 1);
            const T* w = wp + (long)(16 * t + c15) * C + hg * G * 32;
#pragma unroll
            for (int ks = 0; ks < KSO; ++ks) fpj[ks][jj] = *reinterpret_cast<const Frag*>(w + poff(ks));
        }
        __syncthreads();

        // ================= (training) q, k, v of the group -> HBM in the [row, 3C] layout of the unfused kernels
        if (p.qkv) {
            constexpr int CH = 32 / E;                   // 16-byte chunks per (row, part, head)
            T* qg = reinterpret_cast<T*>(p.qkv) + (long)win * 49 * 3 * C + hg * G * 32;
            for (int idx = threadIdx.x; idx < 49 * 3 * G * CH; idx += NT) {
                const int ch = idx % CH, ph = (idx / CH) % (3 * G), r = idx / (CH * 3 * G);
                const int part = ph / G, hl = ph - part * G;
                *reinterpret_cast<Vec*>(qg + (long)r * 3 * C + part * C + hl * 32 + ch * E) =
                    *reinterpret_cast<const Vec*>(qkvt + ((part * G + hl) * 64 + r) * LDH + ch * E);
            }
        }

        // ================= window attention: (head, query tile) units; this wave always runs query tile qt
#pragma unroll
        for (int u = 0; u < UPW; ++u) {
            const int hl = NW == 4 ? u : (wave >> 2) + 2 * u;
            const int h = hg * G + hl;
            const T* Q = qkvt + ((0 * G + hl) * 64) * LDH;
            const T* K = qkvt + ((1 * G + hl) * 64) * LDH;
            const T* V = qkvt + ((2 * G + hl) * 64) * LDH;
            f32x4 sc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) sc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            Frag fq[KBD];
#pragma unroll
            for (int kb = 0; kb < KBD; ++kb) fq[kb] = frag_rowmajor<T>(Q, LDH, 16 * qt, kb * KB);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int kb = 0; kb < KBD; ++kb) M::mma(sc[t], frag_rowmajor<T>(K, LDH, 16 * t, kb * KB), fq[kb]);
            const float* tb = tbl + h * GM::TBL;
            float mx = NEG_BIG;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = fmaf(sc[t][j], p.scale, tb[(lp.ridx[t] >> (8 * j)) & 255u]);
                    if (mbits & (1u << (4 * t + j))) v -= 100.0f;
                    sc[t][j] = v;
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float e = __expf(sc[t][j] - mx); sc[t][j] = e; sum += e; }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.0f / sum;
            if (g == 0 && qrow < 49 && p.lse) p.lse[((long)win * nH + h) * 49 + qrow] = mx + __logf(sum);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) sc[t][j] *= inv;
            f32x4 o[2];
            o[0] = f32x4{0.f, 0.f, 0.f, 0.f}; o[1] = o[0];
#pragma unroll
            for (int kb = 0; kb < KBT; ++kb) {
                const Frag fp = frag_acc<4>(sc, kb, T());
#pragma unroll
                for (int td = 0; td < 2; ++td) M::mma(o[td], frag_tok(V, LDH, 16 * td, kb), fp);
            }
#pragma unroll
            for (int td = 0; td < 2; ++td) store4f(ot + qrow * LDO + hl * 32 + 16 * td + 4 * g, o[td]);
        }
        // the next head group's first qkv weight fragments: requested before the projection slice below
        if (hg + 1 < hg_end) {
#pragma unroll
            for (int jj = 0; jj < NTQW; ++jj) {
                const T* w = wq_ptr(hg + 1, wave + NW * jj);
#pragma unroll
                for (int d = 0; d < PD; ++d) fb[d][jj] = *reinterpret_cast<const Frag*>(w + koff(d));
            }
        }
        __syncthreads();

        // ================= (training) attention output of the group -> HBM, whole rows, for the proj weight gradient
        if (p.ao) {
            constexpr int CH = G * 32 / E;               // 16-byte chunks per row of the group's tile
            T* aog = reinterpret_cast<T*>(p.ao) + (long)win * 49 * C + hg * G * 32;
            for (int idx = threadIdx.x; idx < 49 * CH; idx += NT) {
                const int r = idx / CH, ch = idx - r * CH;
                *reinterpret_cast<Vec*>(aog + (long)r * C + ch * E) = *reinterpret_cast<const Vec*>(ot + r * LDO + ch * E);
            }
        }

        // ================= output projection, this group's k-slice: y[64, C] += O_g[64, 32 G] Wproj[:, 32 G hg ..]^T
        if (!p.head_split) {
            const T* arow = ot + c15 * LDO;
#pragma unroll
            for (int ks = 0; ks < KSO; ++ks) {
                Frag fa[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const Frag*>(arow + 16 * i * LDO + poff(ks));
#pragma unroll
                for (int jj = 0; jj < NTPW; ++jj) {
                    if (wave + NW * jj < NTP) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) M::mma(pacc[i][jj], fpj[ks][jj], fa[i]);
                    }
                }
            }
        }
    }

    // ---- epilogue: + bias, DropPath scale, + shortcut, back to token order
    if (!p.head_split) {
        const float rs = p.rowscale ? p.rowscale[win / p.nW] : 1.0f;
        T* yg = reinterpret_cast<T*>(p.y);
        int tokm[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) tokm[i] = (16 * i + c15 < 49) ? p.w2n[win * 49 + 16 * i + c15] : -1;
#pragma unroll
        for (int jj = 0; jj < NTPW; ++jj) {
            const int t = wave + NW * jj;
            if (t < NTP) {
                const int n = 16 * t + 4 * g;
                const f32x4 b4 = load4f(p.bproj + n);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (tokm[i] >= 0) {
                        const long off = (long)tokm[i] * C + n;
                        const f32x4 r = load4f(xg + off);
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (pacc[i][jj][e] + b4[e]) * rs + r[e];
                        store4f(yg + off, v);
                    }
                }
            }
        }
    }
}

// ===================================================================================================== backward
// mvlt_swin_wmsa_bwd: output-projection dgrad + window-attention backward + qkv dgrad of one block in ONE launch.
//   dao = dy_win Wproj            (per head: [64, C] x [C, 32], weights read through the TRANSPOSED copy wproj_t)
//   per head: P = exp(S - lse), dP = dao V^T, delta = rowsum(P o dP), dS = P o (dP - delta),
//             dQ = scale dS K, dK = scale dS^T Q, dV = P^T dao, dBias[relidx] += dS
//   dxn = dqkv Wqkv               ([64, 96] x [96, C] per head, accumulated over heads; wqkv_t)
// The attention part runs twice, in both orientations (keys on accumulator rows -> dQ, dBias; queries on accumulator
// rows -> dK, dV), like attn.hip, so no score tile is ever transposed through LDS.  q/k/v come from what the forward
// pass saved (qkv_win); dao and delta never leave the chip; dqkv is written once for the qkv weight gradient.
// Workgroups are persistent over windows (<= 512 of them) so the bias-table gradient is flushed once per workgroup.
struct WmsaBwdDev {
    int nwin, nW, res, shift, nH;
    const void* dy; const void* qkv; const float* lse; const void* wproj_t; const void* wqkv_t;
    const float* bias_table; float scale;
    void* dqkv; void* dxn; float* dbias;
};

template <typename T, int C> struct WmsaBwdGeom {
    static constexpr int PAD = sizeof(T) == 2 ? 8 : 4;
    static constexpr int LDX = C + PAD, LDH = 32 + PAD;
    static constexpr size_t XB = (size_t)64 * LDX * sizeof(T);            // dy tile
    static constexpr size_t HB = (size_t)64 * LDH * sizeof(T);            // one [64, 32] head tile
    static constexpr int TBL = 176;
    // dy | q k v | dao | dq dk dv | lse, delta | bias values, bias gradients
    static constexpr size_t bytes(int nH) { return XB + 7 * HB + 2 * 64 * sizeof(float) + (size_t)2 * nH * TBL * sizeof(float); }
};

template <typename T, int C>
__global__ __launch_bounds__(256) void wmsa_bwd_kernel(const WmsaBwdDev p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    using GM = WmsaBwdGeom<T, C>;
    using M = Mma<T>;
    using Frag = typename M::Frag;
    using Vec = typename TypeInfo<T>::Vec;
    constexpr int E = TypeInfo<T>::E, KB = M::KB;
    constexpr int LDX = GM::LDX, LDH = GM::LDH;
    constexpr int KSTEPS = C / KB, KBD = 32 / KB;
    constexpr int TPB = Tok<T>::TPB, KBT = 4 / TPB;
    constexpr int NTP = C / 16, NTPW = (NTP + 3) / 4;
    constexpr int nH = C / 32;
    T* dyt = reinterpret_cast<T*>(smem_raw);
    T* qt = reinterpret_cast<T*>(smem_raw + GM::XB);
    T* kt = qt + 64 * LDH;
    T* vt = kt + 64 * LDH;
    T* dot = vt + 64 * LDH;                 // dao of the head ("dO")
    T* dqt = dot + 64 * LDH;
    T* dkt = dqt + 64 * LDH;
    T* dvt = dkt + 64 * LDH;
    float* lsel = reinterpret_cast<float*>(dvt + 64 * LDH);
    float* dl = lsel + 64;
    float* tbl = dl + 64;
    float* tblg = tbl + nH * GM::TBL;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c15 = lane & 15;
    const T* wp = reinterpret_cast<const T*>(p.wproj_t);
    const T* wq = reinterpret_cast<const T*>(p.wqkv_t);
    const T* dyg = reinterpret_cast<const T*>(p.dy);
    const T* qkvg = reinterpret_cast<const T*>(p.qkv);

    for (int i = threadIdx.x; i < nH * GM::TBL; i += 256) {
        const int h = i / GM::TBL, e = i - h * GM::TBL;
        tbl[i] = e < 169 ? p.bias_table[e * nH + h] : NEG_BIG;
        tblg[i] = 0.f;
    }
    // orientation A: this wave's lanes are queries 16 wave + c15, registers are keys 16 t + 4 g + j
    // orientation B: lanes are keys 16 wave + c15, registers are queries; relidx(q, k) = 168 - relidx(k, q)
    LanePairs lp;
    lp.init(16 * wave + c15, g, p.shift);
    const int lrow = 16 * wave + c15;
    // dBias: every window maps (query, key) to the same lane / register, so dS is summed in registers over all windows
    // this workgroup walks and reaches the LDS table (then global memory) once; per-window LDS atomics -- 4096 colliding
    // ds_add_f32 per head -- were most of the first version's time.  Only for <= 3 heads (stage 0): more heads would need the head loop unrolled (512 registers, spills); they use the LDS table.
    constexpr int REGH = nH <= 3 ? nH : 0;
    f32x4 dbacc[REGH > 0 ? REGH : 1][4];
#pragma unroll
    for (int h = 0; h < (REGH > 0 ? REGH : 1); ++h)
#pragma unroll
        for (int t = 0; t < 4; ++t) dbacc[h][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Software pipeline over (window, head): the dy rows of the next window and the q/k/v rows + lse of the next head
    // are requested into registers one step ahead and written to LDS when their tiles are free, so a workgroup does not
    // start every head with an exposed HBM round trip.
    constexpr int CPR = C / E, CPL = CPR / 4;
    static_assert(CPR % 4 == 0, "row chunks");
    constexpr int CH = 32 / E, NQ = 64 * 3 * CH / 256;          // 16-byte q/k/v chunks per thread and head
    Vec dyr[CPL], qr[NQ];
    float lser = 0.f;
    auto fetch_dy = [&](int win) {
        const int row = threadIdx.x >> 2, sub = threadIdx.x & 3;
        const T* src = dyg + ((long)win * 49 + row) * C;
#pragma unroll
        for (int i = 0; i < CPL; ++i) dyr[i] = row < 49 ? *reinterpret_cast<const Vec*>(src + (sub + 4 * i) * E) : zero_vec<T>();
    };
    auto fetch_qkv = [&](int win, int h) {
        const T* qg = qkvg + (long)win * 49 * 3 * C + h * 32;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int idx = threadIdx.x + 256 * i;
            const int ch = idx % CH, part = (idx / CH) % 3, r = idx / (CH * 3);
            qr[i] = r < 49 ? *reinterpret_cast<const Vec*>(qg + (long)r * 3 * C + part * C + ch * E) : zero_vec<T>();
        }
        if (threadIdx.x < 64) lser = threadIdx.x < 49 ? p.lse[((long)win * nH + h) * 49 + threadIdx.x] : 0.f;
    };
    if ((int)blockIdx.x < p.nwin) { fetch_dy(blockIdx.x); fetch_qkv(blockIdx.x, 0); }

#pragma unroll 1
    for (int win = blockIdx.x; win < p.nwin; win += gridDim.x) {
        __syncthreads();                                       // the previous window is done with every tile
        // ---- dy tile (window order rows are contiguous)
        {
            const int row = threadIdx.x >> 2, sub = threadIdx.x & 3;
#pragma unroll
            for (int i = 0; i < CPL; ++i) *reinterpret_cast<Vec*>(dyt + row * LDX + (sub + 4 * i) * E) = dyr[i];
        }
        uint32_t mbits = 0;
        if (p.shift != 0) {
            const int w = win % p.nW, nwx = p.res / 7;
            const int wy = w / nwx, wx = w - wy * nwx;
            mbits = (wy == nwx - 1 ? lp.rowbits : 0u) | (wx == nwx - 1 ? lp.colbits : 0u);
        }
        f32x4 pacc[4][NTPW];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < NTPW; ++jj) pacc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();

#pragma unroll (nH <= 3 ? nH : 1)
        for (int h = 0; h < nH; ++h) {
            // ================= dao_h rows [16 wave, +16) = dy rows x Wproj[:, 32 h ..]  (both 16-column tiles per wave)
            {
                f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
                const T* arow = dyt + (16 * wave + c15) * LDX + g * E;
                const T* w0 = wp + (long)(32 * h + c15) * C + g * E;
                const T* w1 = w0 + (long)16 * C;
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk) {
                    const Frag fa = *reinterpret_cast<const Frag*>(arow + kk * KB);
                    M::mma(acc[0], *reinterpret_cast<const Frag*>(w0 + kk * KB), fa);
                    M::mma(acc[1], *reinterpret_cast<const Frag*>(w1 + kk * KB), fa);
                }
                store4f(dot + lrow * LDH + 4 * g, acc[0]);
                store4f(dot + lrow * LDH + 16 + 4 * g, acc[1]);
            }
            // ================= q, k, v of the head (saved by the forward pass) and the softmax statistics -> LDS
            {
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
                    const int idx = threadIdx.x + 256 * i;
                    const int ch = idx % CH, part = (idx / CH) % 3, r = idx / (CH * 3);
                    *reinterpret_cast<Vec*>(qt + (part * 64 + r) * LDH + ch * E) = qr[i];
                }
                if (threadIdx.x < 64) lsel[threadIdx.x] = lser;
                // next (window, head) of this workgroup: request it now
                const int nwin_ = h + 1 < nH ? win : win + (int)gridDim.x;
                if (nwin_ < p.nwin) {
                    fetch_qkv(nwin_, h + 1 < nH ? h + 1 : 0);
                    if (h + 1 == nH) fetch_dy(nwin_);
                }
            }
            __syncthreads();
            const float* tb = tbl + h * GM::TBL;
            float* tg = tblg + h * GM::TBL;
            // ================= orientation A: queries on lanes -> delta, dBias, dQ
            {
                f32x4 sc[4], dp[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) { sc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[t] = sc[t]; }
                Frag fq[KBD], fd[KBD];
#pragma unroll
                for (int kb = 0; kb < KBD; ++kb) {
                    fq[kb] = frag_rowmajor<T>(qt, LDH, 16 * wave, kb * KB);
                    fd[kb] = frag_rowmajor<T>(dot, LDH, 16 * wave, kb * KB);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int kb = 0; kb < KBD; ++kb) {
                        M::mma(sc[t], frag_rowmajor<T>(kt, LDH, 16 * t, kb * KB), fq[kb]);
                        M::mma(dp[t], frag_rowmajor<T>(vt, LDH, 16 * t, kb * KB), fd[kb]);
                    }
                const float lq = lsel[lrow];
                float delta = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float v = fmaf(sc[t][j], p.scale, tb[(lp.ridx[t] >> (8 * j)) & 255u]);
                        if (mbits & (1u << (4 * t + j))) v -= 100.0f;
                        const float pr = __expf(v - lq);                  // padded pairs: exp(-1e30 - lse) = 0
                        sc[t][j] = pr;
                        delta += pr * dp[t][j];
                    }
                delta += __shfl_xor(delta, 16, 64);
                delta += __shfl_xor(delta, 32, 64);
                if (g == 0) dl[lrow] = delta;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float ds = sc[t][j] * (dp[t][j] - delta);
                        sc[t][j] = ds;
                        if (REGH > 0) dbacc[REGH > 0 ? h : 0][t][j] += ds;
                        else {
                            const uint32_t ri = (lp.ridx[t] >> (8 * j)) & 255u;
                            if (ri < 169u) atomicAdd(tg + ri, ds);
                        }
                    }
                f32x4 dq[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int kb = 0; kb < KBT; ++kb) {
                    const Frag fp = frag_acc<4>(sc, kb, T());
#pragma unroll
                    for (int td = 0; td < 2; ++td) M::mma(dq[td], frag_tok(kt, LDH, 16 * td, kb), fp);
                }
#pragma unroll
                for (int td = 0; td < 2; ++td) store4f(dqt + lrow * LDH + 16 * td + 4 * g, dq[td] * p.scale);
            }
            __syncthreads();                                   // delta of every query is in LDS
            // ================= orientation B: keys on lanes -> dK, dV
            {
                f32x4 sc[4], dp[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) { sc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[t] = sc[t]; }
                Frag fk[KBD], fv[KBD];
#pragma unroll
                for (int kb = 0; kb < KBD; ++kb) {
                    fk[kb] = frag_rowmajor<T>(kt, LDH, 16 * wave, kb * KB);
                    fv[kb] = frag_rowmajor<T>(vt, LDH, 16 * wave, kb * KB);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int kb = 0; kb < KBD; ++kb) {
                        M::mma(sc[t], frag_rowmajor<T>(qt, LDH, 16 * t, kb * KB), fk[kb]);
                        M::mma(dp[t], frag_rowmajor<T>(dot, LDH, 16 * t, kb * KB), fv[kb]);
                    }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int qr = 16 * t + 4 * g + j;
                        const uint32_t ri = (lp.ridx[t] >> (8 * j)) & 255u;           // relidx(lane key as query, register as key)
                        float v = fmaf(sc[t][j], p.scale, tb[ri < 169u ? 168u - ri : 169u]);
                        if (mbits & (1u << (4 * t + j))) v -= 100.0f;
                        const float pr = __expf(v - lsel[qr]);
                        sc[t][j] = pr;                                                 // P[q, k]
                        dp[t][j] = pr * (dp[t][j] - dl[qr]);                           // dS[q, k]
                    }
                f32x4 dv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, dk[2] = {dv[0], dv[0]};
#pragma unroll
                for (int kb = 0; kb < KBT; ++kb) {
                    const Frag fp = frag_acc<4>(sc, kb, T());
                    const Frag fs = frag_acc<4>(dp, kb, T());
#pragma unroll
                    for (int td = 0; td < 2; ++td) {
                        M::mma(dv[td], frag_tok(dot, LDH, 16 * td, kb), fp);
                        M::mma(dk[td], frag_tok(qt, LDH, 16 * td, kb), fs);
                    }
                }
#pragma unroll
                for (int td = 0; td < 2; ++td) {
                    store4f(dvt + lrow * LDH + 16 * td + 4 * g, dv[td]);
                    store4f(dkt + lrow * LDH + 16 * td + 4 * g, dk[td] * p.scale);
                }
            }
            __syncthreads();
            // ================= dqkv of the head -> HBM (qkv weight gradient), and dxn += dqkv_h Wqkv[(q|k|v) head rows, :]
            {
                T* og = reinterpret_cast<T*>(p.dqkv) + (long)win * 49 * 3 * C + h * 32;
                for (int idx = threadIdx.x; idx < 49 * 3 * CH; idx += 256) {
                    const int ch = idx % CH, part = (idx / CH) % 3, r = idx / (CH * 3);
                    *reinterpret_cast<Vec*>(og + (long)r * 3 * C + part * C + ch * E) =
                        *reinterpret_cast<const Vec*>(dqt + (part * 64 + r) * LDH + ch * E);
                }
#pragma unroll
                for (int part = 0; part < 3; ++part)
#pragma unroll
                    for (int kb = 0; kb < KBD; ++kb) {
                        Frag fa[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) fa[i] = frag_rowmajor<T>(dqt + part * 64 * LDH, LDH, 16 * i, kb * KB);
#pragma unroll
                        for (int jj = 0; jj < NTPW; ++jj) {
                            const int t = wave + 4 * jj;
                            if (t < NTP) {
                                const Frag fb = *reinterpret_cast<const Frag*>(wq + (long)(16 * t + c15) * 3 * C + part * C + h * 32 + kb * KB + g * E);
#pragma unroll
                                for (int i = 0; i < 4; ++i) M::mma(pacc[i][jj], fb, fa[i]);
                            }
                        }
                    }
            }
            // (the next head's first barrier comes after its dao / q,k,v writes, which touch other tiles than dq/dk/dv)
        }
        // ---- dxn rows of this window
        {
            T* xg = reinterpret_cast<T*>(p.dxn) + (long)win * 49 * C;
#pragma unroll
            for (int jj = 0; jj < NTPW; ++jj) {
                const int t = wave + 4 * jj;
                if (t < NTP) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (16 * i + c15 < 49) store4f(xg + (long)(16 * i + c15) * C + 16 * t + 4 * g, pacc[i][jj]);
                }
            }
        }
    }
    __syncthreads();
    if (REGH > 0) {
#pragma unroll
        for (int h = 0; h < (REGH > 0 ? REGH : 1); ++h)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t ri = (lp.ridx[t] >> (8 * j)) & 255u;
                    if (ri < 169u) atomicAdd(tblg + h * GM::TBL + ri, dbacc[h][t][j]);
                }
        __syncthreads();
    }
    if (p.dbias)
        for (int i = threadIdx.x; i < nH * 169; i += 256) {
            const int h = i / 169, e = i - h * 169;
            atomicAdd(&p.dbias[e * nH + h], tblg[h * GM::TBL + e]);
        }
}

template <typename T, int C>
int launch_bwd(const WmsaBwdDev& d, hipStream_t s) {
    using GM = WmsaBwdGeom<T, C>;
    constexpr size_t sh = GM::bytes(C / 32);
    static_assert(sh <= 160 * 1024, "LDS");
    auto k = wmsa_bwd_kernel<T, C>;
    static const hipError_t attr = sh > 64 * 1024
        ? hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) : hipSuccess;
    (void)attr;
    const int grid = d.nwin < 512 ? d.nwin : 512;          // persistent workgroups (512 / 1024 / 2048 measured the same or worse)
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), sh, s, d);
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}

template <typename T, int C, int G, int NW, int MINW = 1>
int launch_fwd(const WmsaDev& d, hipStream_t s) {
    using GM = WmsaGeom<T, C, G>;
    constexpr size_t sh = GM::bytes(C / 32);
    static_assert(sh <= 160 * 1024, "LDS");
    auto k = wmsa_fwd_kernel<T, C, G, NW, MINW>;
    static const hipError_t attr = sh > 64 * 1024            // once per instantiation: the call costs ~10 us of host time
        ? hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) : hipSuccess;
    (void)attr;
    hipLaunchKernelGGL(k, dim3(d.head_split ? d.nwin * (C / 32 / G) : d.nwin), dim3(64 * NW), sh, s, d);
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}

int dispatch_fwd_bf16(const WmsaDev& d, int C, hipStream_t s) {
    using T = bf16_t;
    switch (C) {                                                      // <C, heads per group, waves>: the fastest of the round-2 sweep
        case 96:  return launch_fwd<T, 96, 3, 4>(d, s);               // 3 heads
        case 192: return launch_fwd<T, 192, 6, 8>(d, s);              // 6 heads
        case 384: return launch_fwd<T, 384, 4, 8>(d, s);              // 12 heads
        case 128: return launch_fwd<T, 128, 2, 4>(d, s);              // Swin-B
        case 256: return launch_fwd<T, 256, 2, 4>(d, s);
        case 512: return launch_fwd<T, 512, 4, 8>(d, s);
        default: return MVLT_ERR_UNSUPPORTED;
    }
}

int dispatch_fwd_f32(const WmsaDev& d, int C, hipStream_t s) {
    using T = float;
    switch (C) {
        case 96:  return launch_fwd<T, 96, 1, 4>(d, s);
        case 192: return launch_fwd<T, 192, 1, 4>(d, s);
        case 384: return launch_fwd<T, 384, 1, 4>(d, s);
        case 128: return launch_fwd<T, 128, 1, 4>(d, s);
        case 256: return launch_fwd<T, 256, 1, 4>(d, s);
        default: return MVLT_ERR_UNSUPPORTED;
    }
}

}  // namespace

extern "C" int mvlt_swin_wmsa_supported(int dtype, int C, int nH) {
    if (dtype != MVLT_F32 && dtype != MVLT_BF16) return 0;
    if (nH * 32 != C) return 0;
    if (dtype == MVLT_F32 && C == 512) return 0;                   // f32 tile does not fit the LDS
    return C == 96 || C == 192 || C == 384 || C == 128 || C == 256 || C == 512;
}

extern "C" int mvlt_swin_wmsa_fwd(const MvltSwinWmsa* p, void* stream) {
    MVLT_CHECK(p && p->x && p->w2n && p->ln_gamma && p->ln_beta, MVLT_ERR_ARG);
    MVLT_CHECK(p->wqkv && p->bqkv && p->bias_table, MVLT_ERR_ARG);
    if (p->head_split) MVLT_CHECK(p->attn_out, MVLT_ERR_ARG);
    else MVLT_CHECK(p->y && p->wproj && p->bproj && aligned16(p->y) && aligned16(p->wproj) && aligned16(p->bproj), MVLT_ERR_ARG);
    MVLT_CHECK(p->B > 0 && p->res > 0 && p->res % 7 == 0 && p->shift >= 0 && p->shift < 7, MVLT_ERR_ARG);
    MVLT_CHECK(aligned16(p->x) && aligned16(p->wqkv), MVLT_ERR_ARG);
    MVLT_CHECK(aligned16(p->ln_gamma) && aligned16(p->ln_beta) && aligned16(p->bqkv), MVLT_ERR_ARG);
    MVLT_CHECK((p->mean == nullptr) == (p->rstd == nullptr), MVLT_ERR_ARG);
    if (p->xn_win) MVLT_CHECK(aligned16(p->xn_win), MVLT_ERR_ARG);
    if (p->attn_out) MVLT_CHECK(aligned16(p->attn_out), MVLT_ERR_ARG);
    if (p->qkv_win) MVLT_CHECK(aligned16(p->qkv_win), MVLT_ERR_ARG);
    if (!mvlt_swin_wmsa_supported(p->dtype, p->C, p->nH)) return MVLT_ERR_UNSUPPORTED;
    WmsaDev d{};
    d.nW = (p->res / 7) * (p->res / 7);
    d.nwin = p->B * d.nW; d.res = p->res; d.shift = p->shift; d.nH = p->nH;
    d.x = p->x; d.y = p->y; d.w2n = p->w2n;
    d.gamma = p->ln_gamma; d.beta = p->ln_beta; d.eps = p->ln_eps;
    d.wqkv = p->wqkv; d.bqkv = p->bqkv; d.wproj = p->wproj; d.bproj = p->bproj;
    d.bias_table = p->bias_table; d.scale = p->scale; d.rowscale = p->rowscale;
    d.xn = p->xn_win; d.ao = p->attn_out; d.qkv = p->qkv_win; d.lse = p->lse; d.mean = p->mean; d.rstd = p->rstd;
    d.head_split = p->head_split != 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (p->dtype == MVLT_F32) return dispatch_fwd_f32(d, p->C, s);
    return dispatch_fwd_bf16(d, p->C, s);
}

extern "C" int mvlt_swin_wmsa_bwd_supported(int dtype, int C, int nH) {
    if (nH * 32 != C) return 0;
    if (dtype == MVLT_BF16) return C == 96 || C == 192 || C == 384 || C == 128 || C == 256;
    if (dtype == MVLT_F32) return C == 96 || C == 192 || C == 128;
    return 0;
}

extern "C" int mvlt_swin_wmsa_bwd(const MvltSwinWmsa* p, void* stream) {
    MVLT_CHECK(p && p->dy_win && p->qkv_win && p->lse && p->wproj_t && p->wqkv_t && p->bias_table, MVLT_ERR_ARG);
    MVLT_CHECK(p->dqkv && p->dxn_win, MVLT_ERR_ARG);
    MVLT_CHECK(p->B > 0 && p->res > 0 && p->res % 7 == 0 && p->shift >= 0 && p->shift < 7, MVLT_ERR_ARG);
    MVLT_CHECK(aligned16(p->dy_win) && aligned16(p->qkv_win) && aligned16(p->wproj_t) && aligned16(p->wqkv_t), MVLT_ERR_ARG);
    MVLT_CHECK(aligned16(p->dqkv) && aligned16(p->dxn_win), MVLT_ERR_ARG);
    if (!mvlt_swin_wmsa_bwd_supported(p->dtype, p->C, p->nH)) return MVLT_ERR_UNSUPPORTED;
    WmsaBwdDev d{};
    d.nW = (p->res / 7) * (p->res / 7);
    d.nwin = p->B * d.nW; d.res = p->res; d.shift = p->shift; d.nH = p->nH;
    d.dy = p->dy_win; d.qkv = p->qkv_win; d.lse = p->lse; d.wproj_t = p->wproj_t; d.wqkv_t = p->wqkv_t;
    d.bias_table = p->bias_table; d.scale = p->scale;
    d.dqkv = p->dqkv; d.dxn = p->dxn_win; d.dbias = p->dbias_table;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (p->dtype == MVLT_F32) {
        switch (p->C) {
            case 96:  return launch_bwd<float, 96>(d, s);
            case 192: return launch_bwd<float, 192>(d, s);
            case 128: return launch_bwd<float, 128>(d, s);
            default: return MVLT_ERR_UNSUPPORTED;
        }
    }
    switch (p->C) {
        case 96:  return launch_bwd<bf16_t, 96>(d, s);
        case 192: return launch_bwd<bf16_t, 192>(d, s);
        case 384: return launch_bwd<bf16_t, 384>(d, s);
        case 128: return launch_bwd<bf16_t, 128>(d, s);
        case 256: return launch_bwd<bf16_t, 256>(d, s);
        default: return MVLT_ERR_UNSUPPORTED;
    }
}
