// Fragment helpers shared by the attention kernels (attn.hip) and the fused W-MSA kernels (wmsa.hip).
#pragma once
#include "common.h"

namespace mvlt_attn {

constexpr float NEG_BIG = -1.0e30f;

template <typename T> struct Tok;   // token tiles per MFMA k-block
template <> struct Tok<bf16_t> { static constexpr int TPB = 2; };
template <> struct Tok<float>  { static constexpr int TPB = 1; };

// first-operand fragment: rows = feature d0..d0+15, k-slots = tokens of k-block kb
MVLT_DEV bf16x8 frag_tok(const bf16_t* img, int ld, int d0, int kb) {
    const int l = threadIdx.x & 63;
    const int g = l >> 4, i = l & 15, q = i >> 2, pp = i & 3;
    const bf16_t* p0 = img + (32 * kb + 4 * g + q) * ld + d0 + 4 * pp;
    const bf16_t* p1 = p0 + 16 * ld;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p0);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p1);
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
MVLT_DEV f32x4 frag_tok(const float* img, int ld, int d0, int kb) {
    const int l = threadIdx.x & 63;
    const float* p = img + (16 * kb + 4 * (l >> 4)) * ld + d0 + (l & 15);
    f32x4 r; r[0] = p[0]; r[1] = p[ld]; r[2] = p[2 * ld]; r[3] = p[3 * ld];
    return r;
}
// second-operand fragment from token-tile accumulators
template <int KT> MVLT_DEV bf16x8 frag_acc(const f32x4 (&a)[KT], int kb, bf16_t) {
    bf16x8 r;
    const int t0 = 2 * kb, t1 = 2 * kb + 1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        r[e] = (bf16_t)a[t0][e];
        r[4 + e] = (t1 < KT) ? (bf16_t)a[t1 < KT ? t1 : 0][e] : (bf16_t)0.0f;
    }
    return r;
}
template <int KT> MVLT_DEV f32x4 frag_acc(const f32x4 (&a)[KT], int kb, float) { return a[kb]; }

// v / 7 for 0 <= v < 64 without an integer division (7 * 37 = 259 ~ 2^8): exact on that range
MVLT_DEV int div7(int v) { return (v * 37) >> 8; }
MVLT_DEV int swin_region(int tok, int wy, int wx, int res, int shift) {
    const int ty = div7(tok), tx = tok - 7 * ty;
    const int h = wy * 7 + ty, w = wx * 7 + tx;
    const int bh = h < res - 7 ? 0 : (h < res - shift ? 1 : 2);
    const int bw = w < res - 7 ? 0 : (w < res - shift ? 1 : 2);
    return bh * 3 + bw;
}
MVLT_DEV int rel_index(int q, int k) {
    const int qy = div7(q), ky = div7(k);
    return (qy - ky + 6) * 13 + ((q - 7 * qy) - (k - 7 * ky) + 6);
}

// Window-invariant facts about the 16 (key, query) pairs the score registers of thread tid hold in the keys-on-rows
// orientation (query 16 (tid / 64) + (tid & 15); element e = 4 t + j is key 16 t + 4 ((tid & 63) / 16) + j): the
// relative-position index (169 = outside the 49 x 49 window) and, for the shift of 3 every shifted Swin block uses, the
// pairs that straddle the image border in the last window row / column (low / high half of bits3).  Constant for every
// launch: computing them cost ~400 instructions per thread of every workgroup.
struct SwinPairTable { uint32_t ridx[256][4]; uint32_t bits3[256]; };
constexpr SwinPairTable make_swin_pairs() {
    SwinPairTable r{};
    for (int tid = 0; tid < 256; ++tid) {
        const int q = 16 * (tid >> 6) + (tid & 15), g = (tid & 63) >> 4;
        const int qc = q < 48 ? q : 48, qy = qc / 7, qx = qc % 7;
        uint32_t rowbits = 0, colbits = 0;
        for (int t = 0; t < 4; ++t) {
            uint32_t packed = 0;
            for (int j = 0; j < 4; ++j) {
                const int k = 16 * t + 4 * g + j;
                const bool valid = q < 49 && k < 49;
                const int kc = k < 48 ? k : 48, ky = kc / 7, kx = kc % 7;
                packed |= (uint32_t)(valid ? (qy - ky + 6) * 13 + (qx - kx + 6) : 169) << (8 * j);
                if (valid && ((ky < 4) != (qy < 4))) rowbits |= 1u << (4 * t + j);
                if (valid && ((kx < 4) != (qx < 4))) colbits |= 1u << (4 * t + j);
            }
            r.ridx[tid][t] = packed;
        }
        r.bits3[tid] = rowbits | (colbits << 16);
    }
    return r;
}
static __device__ const SwinPairTable SWIN_PAIRS = make_swin_pairs();

}  // namespace mvlt_attn
