// Shared device helpers for the MVLT gfx950 kernels (CDNA4 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mvlt_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define MVLT_DEV __device__ __forceinline__

#define MVLT_CHECK(cond, code) do { if (!(cond)) return (code); } while (0)
#define MVLT_LAUNCH_CHECK() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return MVLT_ERR_LAUNCH; } while (0)

template <typename T> struct TypeInfo;
template <> struct TypeInfo<float>  { static constexpr int E = 4; using Vec = f32x4;  static constexpr int id = MVLT_F32; };
template <> struct TypeInfo<bf16_t> { static constexpr int E = 8; using Vec = bf16x8; static constexpr int id = MVLT_BF16; };

MVLT_DEV float to_f(float x) { return x; }
MVLT_DEV float to_f(bf16_t x) { return (float)x; }
template <typename T> MVLT_DEV T from_f(float x) { return (T)x; }

// ---------------------------------------------------------------- math
// erf-GELU (nn.GELU default, HF "gelu").  erfc(z), z >= 0, by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, the
// size of an f32 rounding step of the result): branch-free, one v_rcp + one v_exp -- libm's erff costs ~3x as
// many instructions and sits in the epilogue of every FFN-in GEMM and its backward.  exp(-z^2) is returned too:
// the backward needs the Gaussian density exp(-x^2/2) = exp(-z^2) with z = |x|/sqrt(2).
MVLT_DEV float erfc_pos(float z, float& gauss) {
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    gauss = __expf(-z * z);
    return poly * gauss;
}
MVLT_DEV float gelu_f(float x) {
    float gauss;
    const float ec = erfc_pos(fabsf(x) * 0.70710678118654752f, gauss);
    return 0.5f * x * (x >= 0.f ? 2.0f - ec : ec);           // 1 + erf(x / sqrt 2)
}
MVLT_DEV float gelu_grad_f(float x) {
    float gauss;
    const float ec = erfc_pos(fabsf(x) * 0.70710678118654752f, gauss);
    return 0.5f * (x >= 0.f ? 2.0f - ec : ec) + x * 0.3989422804014327f * gauss;
}

// Counter-based dropout RNG: keep(seed, tag, idx) is a pure function so the
// backward pass (and the test oracle, through mvlt_dropout_mask) regenerates
// exactly the forward mask without storing it.
MVLT_DEV uint32_t mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x7feb352dU; h ^= h >> 15; h *= 0x846ca68bU; h ^= h >> 16; return h;
}
MVLT_DEV uint32_t rng_u32(uint64_t seed, uint32_t tag, uint32_t idx) {
    // key depends only on (seed, tag): wave-uniform, hoisted out of element loops by the compiler;
    // per element: one multiply-add + one 32-bit finaliser
    const uint32_t key = mix32((uint32_t)seed ^ (tag * 0x9E3779B9U)) ^ (uint32_t)(seed >> 32);
    return mix32(idx * 0x9E3779B1U + key);
}
// keep with probability (1-p); thresh = p * 2^32
MVLT_DEV bool rng_keep(uint64_t seed, uint32_t tag, uint32_t idx, uint32_t thresh) {
    return rng_u32(seed, tag, idx) >= thresh;
}

// ---------------------------------------------------------------- wave reductions (64 lanes)
MVLT_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
MVLT_DEV float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------- MFMA 16x16 tile abstraction
// One "k-block" = 16 bytes per lane per operand (8 bf16 / 4 f32).  Lane
// l = 16*g + r holds, for row/col r of the 16-wide operand, k-slots
// [g*E, g*E+E) of the k-block.  For bf16 that is the hardware layout of
// v_mfma_f32_16x16x32_bf16 (k = 8g + j).  For f32 the four elements feed four
// v_mfma_f32_16x16x4_f32 (hardware k = g each); the k-slot permutation is the
// same for both operands, so the sum is unchanged.
// Accumulator: c[j] <-> (row = 4*(l>>4) + j of the FIRST operand, col = l&15 of the SECOND).
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    using Frag = bf16x8;
    static constexpr int KB = 32;     // k elements per k-block
    static MVLT_DEV void mma(f32x4& c, const Frag& a, const Frag& b) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    using Frag = f32x4;
    static constexpr int KB = 16;
    static MVLT_DEV void mma(f32x4& c, const Frag& a, const Frag& b) {
#pragma unroll
        for (int i = 0; i < 4; ++i) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], c, 0, 0, 0);
    }
};

// Fragment from an LDS image stored [row][k] (k contiguous), `ld` elements
// per row, plain (un-swizzled) layout.  row0/k0 select the 16 x KB block.
template <typename T>
MVLT_DEV typename Mma<T>::Frag frag_rowmajor(const T* lds, int ld, int row0, int k0) {
    const int l = threadIdx.x & 63;
    const T* p = lds + (row0 + (l & 15)) * ld + k0 + (l >> 4) * TypeInfo<T>::E;
    return *reinterpret_cast<const typename Mma<T>::Frag*>(p);
}

// Fragment from an LDS image stored [k][row] (row contiguous): bf16 uses the
// gfx950 transposed read ds_read_b64_tr_b16 (two 4x16 blocks), f32 four
// scalar reads.  All 64 lanes must be active (EXEC all ones).
MVLT_DEV bf16x8 frag_kmajor(const bf16_t* lds, int ld, int row0, int k0) {
    const int l = threadIdx.x & 63;
    const int g = l >> 4, i = l & 15, q = i >> 2, pp = i & 3;
    const bf16_t* p0 = lds + (k0 + 8 * g + q) * ld + row0 + 4 * pp;
    const bf16_t* p1 = p0 + 4 * ld;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p0);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p1);
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
MVLT_DEV f32x4 frag_kmajor(const float* lds, int ld, int row0, int k0) {
    const int l = threadIdx.x & 63;
    const float* p = lds + (k0 + 4 * (l >> 4)) * ld + row0 + (l & 15);
    f32x4 r;
    r[0] = p[0]; r[1] = p[ld]; r[2] = p[2 * ld]; r[3] = p[3 * ld];
    return r;
}

// ---------------------------------------------------------------- vector IO helpers
template <typename T> MVLT_DEV typename TypeInfo<T>::Vec zero_vec() {
    typename TypeInfo<T>::Vec v;
#pragma unroll
    for (int i = 0; i < TypeInfo<T>::E; ++i) v[i] = (T)0.0f;
    return v;
}

// load 4 consecutive elements as floats (8 B for bf16, 16 B for f32)
MVLT_DEV f32x4 load4f(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
MVLT_DEV f32x4 load4f(const bf16_t* p) {
    bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    f32x4 r; r[0] = (float)v[0]; r[1] = (float)v[1]; r[2] = (float)v[2]; r[3] = (float)v[3];
    return r;
}
MVLT_DEV void store4f(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
MVLT_DEV void store4f(bf16_t* p, const f32x4& v) {
    bf16x4 r; r[0] = (bf16_t)v[0]; r[1] = (bf16_t)v[1]; r[2] = (bf16_t)v[2]; r[3] = (bf16_t)v[3];
    *reinterpret_cast<bf16x4*>(p) = r;
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
