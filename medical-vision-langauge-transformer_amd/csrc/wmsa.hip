// Fused Swin (S)W-MSA for gfx950 (MvltSwinWmsa in include/mvlt_hip.h): the whole attention half of a
// SwinTransformerBlock -- norm1, cyclic shift + window partition, qkv projection, per-head window attention with
// relative-position bias and shift mask, output projection, window reverse + un-shift, DropPath, residual
// (visual_feature_extractor.py:224-254 and :356-384) -- in ONE launch.  The [tokens, 3C] qkv tensor and the
// attention output never make an HBM round trip.
//
// One workgroup (4 waves) owns one 7x7 window = 49 token rows, padded to 64 = four 16-row MFMA tiles:
//   * the window's rows are gathered from token order (row map), LayerNorm-ed with f32 statistics and kept in LDS
//     as the A operand of every projection;
//   * heads are walked in groups of G: the group's 96*G qkv columns are projected with the N tiles dealt to the
//     four waves, so every weight element is loaded exactly once per workgroup, straight from L2 into MFMA
//     B fragments (no LDS staging, no barrier inside the k-loop: the A tile is static);
//   * q, k, v of the group go to a small LDS tile; wave w then runs query tile w of each head with the same
//     in-register transposed-score softmax as attn.hip (bias values come from an LDS copy of the table through
//     window-invariant per-lane indices);
//   * the heads' output tile O[64, 32G] is multiplied into the output projection immediately
//     (y += O_g Wproj[:, g]^T): the projection accumulators live in registers across the head loop, so there is
//     no [64, C] attention-output tile at all;
//   * epilogue: + bias, DropPath scale, + shortcut, scattered back to token order.
// Two barriers per head group.  Training additionally stores what the backward pass needs (window-ordered norm1
// output, attention output, softmax log-sum-exp, LayerNorm statistics) as write-only side outputs.
#include "common.h"
#include "attn_frag.h"

namespace {
using namespace mvlt_attn;

struct WmsaDev {
    int nwin, nW, res, shift, nH;
    const void* x; void* y; const int* w2n;
    const float* gamma; const float* beta; float eps;
    const void* wqkv; const float* bqkv; const void* wproj; const float* bproj;
    const float* bias_table; float scale;
    const float* rowscale;
    void* xn; void* ao; float* lse; float* mean; float* rstd;
};

template <typename T, int C, int G> struct WmsaGeom {
    static constexpr int PAD = sizeof(T) == 2 ? 8 : 4;
    static constexpr int LDX = C + PAD, LDH = 32 + PAD, LDO = G * 32 + PAD;
    static constexpr size_t XB = (size_t)64 * LDX * sizeof(T);
    static constexpr size_t QB = (size_t)3 * G * 64 * LDH * sizeof(T);
    static constexpr size_t OB = (size_t)64 * LDO * sizeof(T);
    static constexpr int TBL = 176;                               // floats per head in the LDS bias table
    static size_t bytes(int nH) { return XB + QB + OB + (size_t)nH * TBL * sizeof(float); }
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

template <typename T, int C, int G>
__global__ __launch_bounds__(256, (C <= 192 ? 2 : 1)) void wmsa_fwd_kernel(const WmsaDev p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    using GM = WmsaGeom<T, C, G>;
    using M = Mma<T>;
    using Frag = typename M::Frag;
    using Vec = typename TypeInfo<T>::Vec;
    constexpr int E = TypeInfo<T>::E, KB = M::KB;
    constexpr int LDX = GM::LDX, LDH = GM::LDH, LDO = GM::LDO;
    constexpr int KSTEPS = C / KB;               // k-steps of the qkv projection
    constexpr int KBD = 32 / KB;                 // k-steps over the head dim
    constexpr int TPB = Tok<T>::TPB;
    constexpr int KBT = 4 / TPB;                 // k-steps over the 64 (padded) keys
    constexpr int NTQ = 6 * G, NTQW = (NTQ + 3) / 4;      // qkv N tiles of a head group, per wave
    constexpr int NTP = C / 16, NTPW = (NTP + 3) / 4;      // proj N tiles, per wave
    constexpr int KSO = G * 32 / KB;             // k-steps of a head group's slice of the output projection
    constexpr int NHG = C / 32 / G;
    static_assert(C % 32 == 0 && (C / 32) % G == 0, "head groups");

    T* xln = reinterpret_cast<T*>(smem_raw);
    T* qkvt = reinterpret_cast<T*>(smem_raw + GM::XB);
    T* ot = reinterpret_cast<T*>(smem_raw + GM::XB + GM::QB);
    float* tbl = reinterpret_cast<float*>(smem_raw + GM::XB + GM::QB + GM::OB);

    const int win = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c15 = lane & 15;
    const T* xg = reinterpret_cast<const T*>(p.x);
    const int nH = C / 32;

    // ---- bias table -> LDS, [head][176] (entries 169.. = -1e30: padded keys / queries)
    for (int i = threadIdx.x; i < nH * GM::TBL; i += 256) {
        const int h = i / GM::TBL, e = i - h * GM::TBL;
        tbl[i] = e < 169 ? p.bias_table[e * nH + h] : NEG_BIG;
    }

    // ---- gather the window's rows (token order -> window order), LayerNorm, normalised tile -> LDS
    {
        constexpr int CPR = C / E, CPL = CPR / 4;         // 16-byte chunks per row / per lane (4 lanes per row)
        static_assert(CPR % 4 == 0, "row chunks");
        const int row = threadIdx.x >> 2, sub = threadIdx.x & 3;
        const bool rv = row < 49;
        const int tok = rv ? p.w2n[win * 49 + row] : 0;
        const T* src = xg + (long)tok * C;
        Vec xv[CPL];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            xv[i] = rv ? *reinterpret_cast<const Vec*>(src + (sub + 4 * i) * E) : zero_vec<T>();
#pragma unroll
            for (int e = 0; e < E; ++e) s += to_f(xv[i][e]);
        }
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64);
        const float mean = s / C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < CPL; ++i)
#pragma unroll
            for (int e = 0; e < E; ++e) { const float d = to_f(xv[i][e]) - mean; q += d * d; }
        q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64);
        const float rstd = rsqrtf(q / C + p.eps);
        if (rv && sub == 0 && p.mean) { p.mean[tok] = mean; p.rstd[tok] = rstd; }
        T* xs = p.xn ? reinterpret_cast<T*>(p.xn) + ((long)win * 49 + row) * C : nullptr;
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int c = (sub + 4 * i) * E;
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
    LanePairs lp;
    lp.init(16 * wave + c15, g, p.shift);
    uint32_t mbits = 0;
    if (p.shift != 0) {
        const int w = win % p.nW, nwx = p.res / 7;
        const int wy = w / nwx, wx = w - wy * nwx;
        mbits = (wy == nwx - 1 ? lp.rowbits : 0u) | (wx == nwx - 1 ? lp.colbits : 0u);
    }
    const int qrow = 16 * wave + c15;

    f32x4 pacc[4][NTPW];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < NTPW; ++jj) pacc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};

    const T* wq = reinterpret_cast<const T*>(p.wqkv);
    const T* wp = reinterpret_cast<const T*>(p.wproj);
    __syncthreads();

#pragma unroll 1
    for (int hg = 0; hg < NHG; ++hg) {
        // ================= qkv projection of head group hg: [64, C] x [C, 96 G]
        {
            f32x4 acc[4][NTQW];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < NTQW; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
            const T* wrow[NTQW];
#pragma unroll
            for (int jj = 0; jj < NTQW; ++jj) {
                const int t = min(wave + 4 * jj, NTQ - 1);
                const int part = t / (2 * G), within = t - part * 2 * G;
                wrow[jj] = wq + (long)(part * C + hg * G * 32 + within * 16 + c15) * C + g * E;
            }
            const T* arow = xln + c15 * LDX + g * E;
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                Frag fa[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const Frag*>(arow + 16 * i * LDX + kk * KB);
#pragma unroll
                for (int jj = 0; jj < NTQW; ++jj) {
                    if (wave + 4 * jj < NTQ) {
                        const Frag fb = *reinterpret_cast<const Frag*>(wrow[jj] + kk * KB);
#pragma unroll
                        for (int i = 0; i < 4; ++i) M::mma(acc[i][jj], fb, fa[i]);
                    }
                }
            }
            // + bias, to the q/k/v LDS tiles [part][head][row][32]
#pragma unroll
            for (int jj = 0; jj < NTQW; ++jj) {
                const int t = wave + 4 * jj;
                if (t < NTQ) {
                    const int part = t / (2 * G), within = t - part * 2 * G;
                    const f32x4 b4 = load4f(p.bqkv + part * C + hg * G * 32 + within * 16 + 4 * g);
                    T* dst = qkvt + ((part * G + (within >> 1)) * 64 + c15) * LDH + (within & 1) * 16 + 4 * g;
#pragma unroll
                    for (int i = 0; i < 4; ++i) store4f(dst + 16 * i * LDH, acc[i][jj] + b4);
                }
            }
        }
        __syncthreads();

        // ================= window attention: wave w = query tile w of each head of the group
#pragma unroll
        for (int hl = 0; hl < G; ++hl) {
            const int h = hg * G + hl;
            const T* Q = qkvt + ((0 * G + hl) * 64) * LDH;
            const T* K = qkvt + ((1 * G + hl) * 64) * LDH;
            const T* V = qkvt + ((2 * G + hl) * 64) * LDH;
            f32x4 sc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) sc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            Frag fq[KBD];
#pragma unroll
            for (int kb = 0; kb < KBD; ++kb) fq[kb] = frag_rowmajor<T>(Q, LDH, 16 * wave, kb * KB);
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
        __syncthreads();

        // ================= (training) attention output of the group -> HBM, whole rows, for the proj weight gradient
        if (p.ao) {
            constexpr int CH = G * 32 / E;               // 16-byte chunks per row of the group's tile
            T* aog = reinterpret_cast<T*>(p.ao) + (long)win * 49 * C + hg * G * 32;
            for (int idx = threadIdx.x; idx < 49 * CH; idx += 256) {
                const int r = idx / CH, ch = idx - r * CH;
                *reinterpret_cast<Vec*>(aog + (long)r * C + ch * E) = *reinterpret_cast<const Vec*>(ot + r * LDO + ch * E);
            }
        }

        // ================= output projection, this group's k-slice: y[64, C] += O_g[64, 32 G] Wproj[:, 32 G hg ..]^T
        {
            const T* arow = ot + c15 * LDO + g * E;
#pragma unroll
            for (int ks = 0; ks < KSO; ++ks) {
                Frag fa[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const Frag*>(arow + 16 * i * LDO + ks * KB);
#pragma unroll
                for (int jj = 0; jj < NTPW; ++jj) {
                    const int t = wave + 4 * jj;
                    if (t < NTP) {
                        const Frag fb = *reinterpret_cast<const Frag*>(wp + (long)(16 * t + c15) * C + hg * G * 32 + ks * KB + g * E);
#pragma unroll
                        for (int i = 0; i < 4; ++i) M::mma(pacc[i][jj], fb, fa[i]);
                    }
                }
            }
        }
    }

    // ---- epilogue: + bias, DropPath scale, + shortcut, back to token order
    {
        const float rs = p.rowscale ? p.rowscale[win / p.nW] : 1.0f;
        T* yg = reinterpret_cast<T*>(p.y);
        int tokm[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) tokm[i] = (16 * i + c15 < 49) ? p.w2n[win * 49 + 16 * i + c15] : -1;
#pragma unroll
        for (int jj = 0; jj < NTPW; ++jj) {
            const int t = wave + 4 * jj;
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

template <typename T, int C, int G>
int launch_fwd(const WmsaDev& d, hipStream_t s) {
    using GM = WmsaGeom<T, C, G>;
    const size_t sh = GM::bytes(C / 32);
    if (sh > 160 * 1024) return MVLT_ERR_UNSUPPORTED;
    auto k = wmsa_fwd_kernel<T, C, G>;
    if (sh > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL(k, dim3(d.nwin), dim3(256), sh, s, d);
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}

template <typename T>
int dispatch_fwd(const WmsaDev& d, int C, hipStream_t s) {
    constexpr bool B16 = sizeof(T) == 2;
    switch (C) {
        case 96:  return launch_fwd<T, 96, 1>(d, s);                 // 3 heads
        case 192: return launch_fwd<T, 192, (B16 ? 2 : 1)>(d, s);
        case 384: return launch_fwd<T, 384, (B16 ? 2 : 1)>(d, s);
        case 128: return launch_fwd<T, 128, (B16 ? 2 : 1)>(d, s);    // Swin-B
        case 256: return launch_fwd<T, 256, (B16 ? 2 : 1)>(d, s);
        case 512: return launch_fwd<T, 512, (B16 ? 2 : 1)>(d, s);
        default: return MVLT_ERR_UNSUPPORTED;
    }
}

}  // namespace

extern "C" int mvlt_swin_wmsa_supported(int dtype, int C, int nH) {
    if (dtype != MVLT_F32 && dtype != MVLT_BF16) return 0;
    if (nH * 32 != C) return 0;
    return C == 96 || C == 192 || C == 384 || C == 128 || C == 256 || C == 512;
}

extern "C" int mvlt_swin_wmsa_fwd(const MvltSwinWmsa* p, void* stream) {
    MVLT_CHECK(p && p->x && p->y && p->w2n && p->ln_gamma && p->ln_beta, MVLT_ERR_ARG);
    MVLT_CHECK(p->wqkv && p->bqkv && p->wproj && p->bproj && p->bias_table, MVLT_ERR_ARG);
    MVLT_CHECK(p->B > 0 && p->res > 0 && p->res % 7 == 0 && p->shift >= 0 && p->shift < 7, MVLT_ERR_ARG);
    MVLT_CHECK(aligned16(p->x) && aligned16(p->y) && aligned16(p->wqkv) && aligned16(p->wproj), MVLT_ERR_ARG);
    MVLT_CHECK(aligned16(p->ln_gamma) && aligned16(p->ln_beta) && aligned16(p->bqkv) && aligned16(p->bproj), MVLT_ERR_ARG);
    MVLT_CHECK((p->mean == nullptr) == (p->rstd == nullptr), MVLT_ERR_ARG);
    if (p->xn_win) MVLT_CHECK(aligned16(p->xn_win), MVLT_ERR_ARG);
    if (p->attn_out) MVLT_CHECK(aligned16(p->attn_out), MVLT_ERR_ARG);
    if (!mvlt_swin_wmsa_supported(p->dtype, p->C, p->nH)) return MVLT_ERR_UNSUPPORTED;
    WmsaDev d{};
    d.nW = (p->res / 7) * (p->res / 7);
    d.nwin = p->B * d.nW; d.res = p->res; d.shift = p->shift; d.nH = p->nH;
    d.x = p->x; d.y = p->y; d.w2n = p->w2n;
    d.gamma = p->ln_gamma; d.beta = p->ln_beta; d.eps = p->ln_eps;
    d.wqkv = p->wqkv; d.bqkv = p->bqkv; d.wproj = p->wproj; d.bproj = p->bproj;
    d.bias_table = p->bias_table; d.scale = p->scale; d.rowscale = p->rowscale;
    d.xn = p->xn_win; d.ao = p->attn_out; d.lse = p->lse; d.mean = p->mean; d.rstd = p->rstd;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (p->dtype == MVLT_F32) return dispatch_fwd<float>(d, p->C, s);
    return dispatch_fwd<bf16_t>(d, p->C, s);
}

extern "C" int mvlt_swin_wmsa_bwd(const MvltSwinWmsa* p, void* stream) {
    (void)p; (void)stream;
    return MVLT_ERR_UNSUPPORTED;
}
