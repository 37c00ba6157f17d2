// LayerNorm forward / backward for gfx950 (HBM-bound; see MvltLayerNorm in
// include/mvlt_hip.h).  A row is owned by LPR lanes of one wave (16/32/64) and
// kept in registers as NV 4-element vectors per lane (8/16-byte coalesced
// loads), so x is read exactly once: algorithmic traffic = read x + write y.
// The store can scatter rows (window partition + cyclic shift), the load can
// gather the PatchMerging 2x2 neighbourhood, and GELU can be fused.
#include "common.h"
#include <cstdlib>

namespace {
constexpr int LN_PARTS_STRIDE = 512;   // == LN_BWD_PARTS: row stride between the dgamma and dbeta partial blocks

struct LnDev {
    int rows, C; float eps;
    const void* x; const float* gamma; const float* beta;
    void* y; void* y_pre; float* mean; float* rstd;
    const int* rowmap; int mH, mW; int gelu;
    // backward
    const void* dy; const void* dres; void* dx;
    float* part_g; float* part_b; int nparts;
    // optional second output: dz[zmap ? zmap[r] : r] = dropmask(dx[r]) * zscale[r / zrps]
    void* dz; const int* zmap; const float* zscale; int zrps; uint32_t zthresh; float zdscale; uint64_t seed; uint32_t tag;
    const int* rows_dev;     // optional: valid rows on the device (ragged batches planned on the GPU)
    int dy_parts; long dy_part_bytes;      // dy = sum of dy_parts tensors dy_part_bytes apart (MvltLayerNormBwd.dy_parts), or 0 / 1
};
MVLT_DEV int ln_rows(const LnDev& p) { return p.rows_dev ? min(p.rows, max(*p.rows_dev, 0)) : p.rows; }

// sum over the LPR consecutive lanes of a row: the steps inside a 16-lane DPP row are DPP adds (no LDS round trip: the
// reduction sits on every row's critical path), the 16 / 32 steps go through the permute network
template <int LPR> MVLT_DEV float group_sum(float v) {
    if (LPR >= 2) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    if (LPR >= 4) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    if (LPR >= 8) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
    if (LPR >= 16) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true)); // row_mirror
    if (LPR >= 32) v += __shfl_xor(v, 16, 64);
    if (LPR >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}

// element offset of column c of logical row r
template <bool MERGE>
MVLT_DEV long in_offset(const LnDev& p, int r, int c) {
    if (!MERGE) return (long)r * p.C + c;
    const int Cq = p.C >> 2, Ho = p.mH >> 1, Wo = p.mW >> 1;
    const int b = r / (Ho * Wo), rem = r % (Ho * Wo), i = rem / Wo, j = rem % Wo;
    const int s = c / Cq, within = c % Cq;
    const int h = 2 * i + (s & 1), w = 2 * j + (s >> 1);
    return ((long)(b * p.mH + h) * p.mW + w) * Cq + within;
}

template <typename T, int LPR, int NV, bool MERGE>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const LnDev p) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sl = lane % LPR;
    const int r = (blockIdx.x * 4 + wave) * RPW + lane / LPR;
    const bool rv = r < ln_rows(p);
    const T* x = reinterpret_cast<const T*>(p.x);
    f32x4 v[NV], gm[NV], bt[NV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = 4 * (sl + LPR * j);
        v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (rv && c < p.C) v[j] = load4f(x + in_offset<MERGE>(p, r, c));
    }
    // gamma / beta are loaded HERE, with the row: vmcnt completes in order, so a load issued between the stores below
    // would wait for the store in front of it
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = min(4 * (sl + LPR * j), p.C - 4);
        gm[j] = load4f(p.gamma + c); bt[j] = load4f(p.beta + c);
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) s += v[j][0] + v[j][1] + v[j][2] + v[j][3];
    const float mean = group_sum<LPR>(s) / p.C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = 4 * (sl + LPR * j);
        if (c < p.C) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[j][e] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(group_sum<LPR>(q) / p.C + p.eps);
    if (!rv) return;
    if (sl == 0) {
        if (p.mean) p.mean[r] = mean;
        if (p.rstd) p.rstd[r] = rstd;
    }
    const int ro = p.rowmap ? p.rowmap[r] : r;
    T* y = reinterpret_cast<T*>(p.y) + (long)ro * p.C;
    T* yp = p.y_pre ? reinterpret_cast<T*>(p.y_pre) + (long)ro * p.C : nullptr;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = 4 * (sl + LPR * j);
        if (c < p.C) {
            const f32x4 g = gm[j], b = bt[j];
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[j][e] - mean) * rstd * g[e] + b[e];
            if (p.gelu) {
                if (yp) store4f(yp + c, o);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = gelu_f(o[e]);
            }
            store4f(y + c, o);
        }
    }
}

template <typename T, int LPR, int NV, bool MERGE>
__global__ __launch_bounds__(1024) void ln_bwd_kernel(const LnDev p) {
    constexpr int RPW = 64 / LPR;
    extern __shared__ __attribute__((aligned(16))) float red[];   // [2][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwave = blockDim.x >> 6;
    const int sl = lane % LPR;
    const T* x = reinterpret_cast<const T*>(p.x);
    const T* dy = reinterpret_cast<const T*>(p.dy);
    const T* ypre = reinterpret_cast<const T*>(p.y_pre);
    const T* dres = reinterpret_cast<const T*>(p.dres);
    T* dx = reinterpret_cast<T*>(p.dx);
    f32x4 ag[NV], ab[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) { ag[j] = f32x4{0.f, 0.f, 0.f, 0.f}; ab[j] = ag[j]; }
    const float invC = 1.0f / p.C;
    const int nrows = ln_rows(p);
    f32x4 gmv[NV];                                   // gamma: the same for every row
#pragma unroll
    for (int j = 0; j < NV; ++j) gmv[j] = load4f(p.gamma + min(4 * (sl + LPR * j), p.C - 4));
    for (int r0 = (blockIdx.x * nwave + wave) * RPW; r0 < nrows; r0 += gridDim.x * nwave * RPW) {
        const int r = r0 + lane / LPR;
        const bool rv = r < nrows;
        const float mean = rv ? p.mean[r] : 0.f, rstd = rv ? p.rstd[r] : 0.f;
        const int rd = (rv && p.rowmap) ? p.rowmap[r] : r;
        f32x4 xh[NV], g[NV], rs[NV];
        float s1 = 0.f, s2 = 0.f;
        // the residual-path gradient is loaded with the row (a load issued between the stores below would wait for the
        // store in front of it: vmcnt completes in order)
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = 4 * (sl + LPR * j);
            rs[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (dres && rv && c < p.C) rs[j] = load4f(dres + in_offset<MERGE>(p, r, c));
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = 4 * (sl + LPR * j);
            xh[j] = f32x4{0.f, 0.f, 0.f, 0.f}; g[j] = xh[j];
            if (rv && c < p.C) {
                f32x4 xv = load4f(x + in_offset<MERGE>(p, r, c));
                f32x4 d = load4f(dy + (long)rd * p.C + c);
                if (p.gelu) {
                    f32x4 yp = load4f(ypre + (long)rd * p.C + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[e] *= gelu_grad_f(yp[e]);
                }
                const f32x4 gm = gmv[j];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[j][e] = (xv[e] - mean) * rstd;
                    g[j][e] = d[e] * gm[e];
                    s1 += g[j][e]; s2 += g[j][e] * xh[j][e];
                    ag[j][e] += d[e] * xh[j][e]; ab[j][e] += d[e];
                }
            }
        }
        s1 = group_sum<LPR>(s1) * invC;
        s2 = group_sum<LPR>(s2) * invC;
        if (rv) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int c = 4 * (sl + LPR * j);
                if (c < p.C) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = rstd * (g[j][e] - s1 - xh[j][e] * s2);
                    const long off = in_offset<MERGE>(p, r, c);
                    o += rs[j];
                    store4f(dx + off, o);
                    if (!MERGE && p.dz) {           // branch gradient for the residual branch that consumed this tensor
                        if (p.zthresh) {
                            const uint32_t base = (uint32_t)r * (uint32_t)p.C + c;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = rng_keep(p.seed, p.tag, base + e, p.zthresh) ? o[e] * p.zdscale : 0.f;
                        }
                        if (p.zscale) { const float zs = p.zscale[r / p.zrps];
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] *= zs; }
                        const int zr = p.zmap ? p.zmap[r] : r;
                        store4f(reinterpret_cast<T*>(p.dz) + (long)zr * p.C + c, o);
                    }
                }
            }
        }
    }
    // block-level reduction without atomics: lanes of different row groups (LPR < 64) fold by
    // xor-shuffle, every wave stores its partial row to LDS [wave][2][C], then a conflict-free
    // column sum over the waves.
#pragma unroll
    for (int j = 0; j < NV; ++j) {
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { ag[j][e] += __shfl_xor(ag[j][e], o, 64); ab[j][e] += __shfl_xor(ab[j][e], o, 64); }
        }
        const int c = 4 * (sl + LPR * j);
        if (c < p.C && lane < LPR) {
            *reinterpret_cast<f32x4*>(&red[(wave * 2 + 0) * p.C + c]) = ag[j];
            *reinterpret_cast<f32x4*>(&red[(wave * 2 + 1) * p.C + c]) = ab[j];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * p.C; c += blockDim.x) {
        const int which = c >= p.C, cc = which ? c - p.C : c;
        float sum = 0.f;
        for (int w = 0; w < nwave; ++w) sum += red[(w * 2 + which) * p.C + cc];
        (which ? p.part_b : p.part_g)[(long)blockIdx.x * p.C + cc] = sum;
    }
}

// ---- LayerNorm backward, low-footprint form (round 6; profiles/r6_ln_bwd.md) --------------------------------------------------
// Inside the training step this kernel runs on the dgrad chain BESIDE a layer's weight-gradient workgroups (side stream), which hold
// 336-496 of a SIMD's 512 VGPRs and 96-128 KB of a CU's LDS.  The kernel above is compiled for 128 VGPRs per wave and launched as
// 8-wave blocks with 25-49 KB of LDS: beside two weight-gradient workgroups NOT ONE such block fits on a CU, so the launch runs on
// the CUs the weight gradients left half empty -- 38 us in the step against 9 us alone for the BertLayer shape
// (scripts/in_situ_overlap.py on a kernel trace; counters cannot show it, `rocprofv3 --pmc` serialises dispatches).
// This form is built to fit INTO what is left: 4-wave blocks, <= 80 VGPRs (rows stay packed as loaded and are converted where
// they are used, twice; the next row group's loads are issued before the current one is reduced, so a wave keeps two row
// groups in flight), 2 x C floats of LDS (the waves of a block fold their parameter-gradient rows one after the other: fixed
// order, no atomics).  Same arithmetic, same partial-row contract (one partial row pair per block) as the kernel above.
MVLT_DEV f32x4 ln_unpack4(const u32x2& w) {
    f32x4 r;
    r[0] = __builtin_bit_cast(float, w[0] << 16); r[1] = __builtin_bit_cast(float, w[0] & 0xffff0000u);
    r[2] = __builtin_bit_cast(float, w[1] << 16); r[3] = __builtin_bit_cast(float, w[1] & 0xffff0000u);
    return r;
}
MVLT_DEV u32x2 ln_pack4(const f32x4& v) {
    bf16x4 r; r[0] = (bf16_t)v[0]; r[1] = (bf16_t)v[1]; r[2] = (bf16_t)v[2]; r[3] = (bf16_t)v[3];
    return __builtin_bit_cast(u32x2, r);
}

// bf16, C == 4 LPR NV exactly (every width of configs #2 / #5), no GELU, no patch-merging gather.  All row operands go through
// buffer descriptors: one 32-bit offset register per row instead of a 64-bit pointer per operand, rows past the end read
// zeros and their stores are dropped by the range check -- no clamps, no predicated loads.  PF: two row groups in flight.
struct LnRowSet { uint32_t voff, zoff; float mean, rstd, zs; };
// PARTS: dy is the SUM of up to four tensors (the partial qkv-dgrad products of mvlt_swin_wmsa2_bwd): the extra parts are loaded
// beside the first one -- absent parts through an out-of-range offset, which a buffer load answers with zeros -- and added in f32
// where the row's values are first used; the sum is rounded to bf16 once (12 more registers per row group: looser launch bounds).
template <int LPR, int NV, bool PF, bool PARTS = false>
__global__ __launch_bounds__(256, PARTS ? 4 : (PF ? (NV <= 2 ? 5 : 4) : (NV <= 3 ? 6 : 5))) void ln_bwd2_kernel(const LnDev p) {
    constexpr int RPW = 64 / LPR;
    constexpr int JB = LPR * 8;                                       // bytes between a lane's 4-element chunks
    constexpr uint32_t OOB = 0x7fffffffu;
    extern __shared__ __attribute__((aligned(16))) float red[];      // [2][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sl = lane % LPR;
    const int C = p.C, rowb = C * 2;
    const int nrows = ln_rows(p);
    const int stride = gridDim.x * 4 * RPW;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, nrows * rowb, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy), 0, PARTS ? (int)((p.dy_parts - 1) * p.dy_part_bytes) + p.rows * rowb : p.rows * rowb, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dres ? p.dres : p.x), 0, p.dres ? nrows * rowb : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdx = __builtin_amdgcn_make_buffer_rsrc(p.dx, 0, nrows * rowb, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdz = __builtin_amdgcn_make_buffer_rsrc(p.dz ? p.dz : p.dx, 0, p.dz ? p.rows * rowb : 0, 0x00020000);
    f32x4 ag[NV], ab[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) { ag[j] = f32x4{0.f, 0.f, 0.f, 0.f}; ab[j] = ag[j]; }
    // gamma lives in LDS behind the reduction rows (12 registers less per wave at C = 768: the difference between one and two
    // resident blocks per CU beside the weight-gradient workgroups); re-read per row group through an address hipcc cannot hoist
    float* lgam = red + 2 * C;
    for (int c = threadIdx.x; c < C; c += 256) lgam[c] = p.gamma[c];
    __syncthreads();
    uint32_t gaddr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)lgam + sl * 16;
    auto gam = [&](int j) -> f32x4 {
        asm volatile("" : "+v"(gaddr));
        return *reinterpret_cast<__attribute__((address_space(3))) f32x4*>(gaddr + j * LPR * 16);
    };

    u32x2 dpx[PARTS ? 3 : 1][NV];                                     // PARTS: parts 1 .. 3 of the row group in flight
    auto request = [&](int r0, LnRowSet& st, u32x2 (&xr)[NV], u32x2 (&dr)[NV], u32x2 (&rr)[NV]) {
        const int r = r0 + lane / LPR, rc = min(r, nrows - 1);
        const bool rv = r < nrows;
        st.mean = p.mean[rc]; st.rstd = p.rstd[rc];
        const int rd = p.rowmap ? p.rowmap[rc] : rc;
        st.zs = p.zscale ? p.zscale[rc / p.zrps] : 1.0f;
        const int zr = p.zmap ? p.zmap[rc] : rc;
        st.voff = (uint32_t)(r * rowb + sl * 8);
        st.zoff = rv ? (uint32_t)(zr * rowb + sl * 8) : OOB;
        const uint32_t doff = rv ? (uint32_t)(rd * rowb + sl * 8) : OOB;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            xr[j] = __builtin_amdgcn_raw_buffer_load_b64(rx, st.voff + j * JB, 0, 0);
            dr[j] = __builtin_amdgcn_raw_buffer_load_b64(rdy, doff + j * JB, 0, 0);
        }
        if constexpr (PARTS) {
#pragma unroll
            for (int pp = 0; pp < 3; ++pp) {
                const uint32_t po = (rv && pp + 1 < p.dy_parts) ? doff + (uint32_t)((pp + 1) * p.dy_part_bytes) : OOB;
#pragma unroll
                for (int j = 0; j < NV; ++j) dpx[pp][j] = __builtin_amdgcn_raw_buffer_load_b64(rdy, po + j * JB, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) rr[j] = __builtin_amdgcn_raw_buffer_load_b64(rdr, st.voff + j * JB, 0, 0);
    };
    // (the packed registers pass through an opaque asm in front of each use: hipcc otherwise keeps the f32 forms of pass 1 alive
    // for pass 2 -- 36 more registers per row group than the packed operands it was given)
    auto opaque = [](u32x2& v) { asm volatile("" : "+v"(v)); };
    auto finish = [&](const LnRowSet& st, u32x2 (&xr)[NV], u32x2 (&dr)[NV], u32x2 (&rr)[NV]) {
        float s1 = 0.f, s2 = 0.f;
        const float nm = -st.mean * st.rstd;
        if constexpr (PARTS) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                f32x4 d = ln_unpack4(dr[j]);
#pragma unroll
                for (int pp = 0; pp < 3; ++pp) d += ln_unpack4(dpx[pp][j]);
                dr[j] = ln_pack4(d);
            }
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            opaque(xr[j]); opaque(dr[j]);
            const f32x4 xv = ln_unpack4(xr[j]), d = ln_unpack4(dr[j]), gm = gam(j);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = fmaf(xv[e], st.rstd, nm), g = d[e] * gm[e];
                s1 += g; s2 = fmaf(g, xh, s2);
                ag[j][e] = fmaf(d[e], xh, ag[j][e]); ab[j][e] += d[e];
            }
        }
        // (the parameter-gradient sums are complete HERE: left alone, hipcc moves them behind the stores below and carries the
        // 8 NV products across the whole second pass)
#pragma unroll
        for (int j = 0; j < NV; ++j) asm volatile("" : "+v"(ag[j]), "+v"(ab[j]));
        s1 = group_sum<LPR>(s1) * (1.0f / (4 * LPR * NV));
        s2 = group_sum<LPR>(s2) * (1.0f / (4 * LPR * NV));
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            opaque(xr[j]); opaque(dr[j]); opaque(rr[j]);
            const f32x4 xv = ln_unpack4(xr[j]), d = ln_unpack4(dr[j]), rs = ln_unpack4(rr[j]), gm = gam(j);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = fmaf(xv[e], st.rstd, nm);
                o[e] = fmaf(st.rstd, fmaf(d[e], gm[e], -s1) - xh * s2, rs[e]);
            }
            __builtin_amdgcn_raw_buffer_store_b64(ln_pack4(o), rdx, st.voff + j * JB, 0, 0);
            if (p.dz) {                                         // gradient entering the residual branch that produced this tensor
                if (p.zthresh) {
                    const uint32_t base = (st.voff >> 1) + 4 * LPR * j;          // element index r C + c
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = rng_keep(p.seed, p.tag, base + e, p.zthresh) ? o[e] * p.zdscale : 0.f;
                }
                o *= st.zs;
                __builtin_amdgcn_raw_buffer_store_b64(ln_pack4(o), rdz, st.zoff + j * JB, 0, 0);
            }
        }
    };
    LnRowSet sa, sb;
    u32x2 xa[NV], da[NV], ra[NV], xb[NV], db[NV], rb[NV];
    int r0 = (blockIdx.x * 4 + wave) * RPW;
    if constexpr (PF) {
        if (r0 < nrows) request(r0, sa, xa, da, ra);
        while (r0 < nrows) {
            const int r1 = r0 + stride;
            if (r1 < nrows) request(r1, sb, xb, db, rb);
            finish(sa, xa, da, ra);
            if (r1 >= nrows) break;
            const int r2 = r1 + stride;
            if (r2 < nrows) request(r2, sa, xa, da, ra);
            finish(sb, xb, db, rb);
            r0 = r2;
        }
    } else {
        for (; r0 < nrows; r0 += stride) { request(r0, sa, xa, da, ra); finish(sa, xa, da, ra); }
    }
    // the row groups of a wave (LPR < 64) fold by xor-shuffle; the waves of the block add their rows into LDS one after the other
#pragma unroll
    for (int j = 0; j < NV; ++j) {
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { ag[j][e] += __shfl_xor(ag[j][e], o, 64); ab[j][e] += __shfl_xor(ab[j][e], o, 64); }
        }
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (wave == w && lane < LPR) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                f32x4* g = reinterpret_cast<f32x4*>(&red[4 * (sl + LPR * j)]);
                f32x4* b = reinterpret_cast<f32x4*>(&red[C + 4 * (sl + LPR * j)]);
                if (w == 0) { *g = ag[j]; *b = ab[j]; } else { *g += ag[j]; *b += ab[j]; }
            }
        }
        __syncthreads();
    }
    for (int c = threadIdx.x; c < 2 * C; c += 256)
        (c >= C ? p.part_b : p.part_g)[(long)blockIdx.x * C + (c >= C ? c - C : c)] = red[c];
}

// partial rows -> dgamma/dbeta: block = 64 columns x 16 row groups (one wave each), coalesced 256-B row reads
__global__ __launch_bounds__(1024) void ln_param_reduce_kernel(const float* part_g, const float* part_b, int nparts, int C,
                                                               float* dgamma, float* dbeta, int accumulate) {
    __shared__ float red[2][16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float g = 0.f, b = 0.f;
    if (c < C) {
        float g1 = 0.f, g2 = 0.f, g3 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;      // four partial rows in flight per step
        int i = w;
        for (; i + 48 < nparts; i += 64) {
            const long o = (long)i * C + c, s16 = 16L * C;
            g += part_g[o]; g1 += part_g[o + s16]; g2 += part_g[o + 2 * s16]; g3 += part_g[o + 3 * s16];
            b += part_b[o]; b1 += part_b[o + s16]; b2 += part_b[o + 2 * s16]; b3 += part_b[o + 3 * s16];
        }
        for (; i < nparts; i += 16) { g += part_g[(long)i * C + c]; b += part_b[(long)i * C + c]; }
        g += g1 + g2 + g3; b += b1 + b2 + b3;
    }
    red[0][w][lane] = g; red[1][w][lane] = b;
    __syncthreads();
    if (w == 0 && c < C) {
        g = 0.f; b = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { g += red[0][i][lane]; b += red[1][i][lane]; }
        if (accumulate) { g += dgamma[c]; b += dbeta[c]; }
        dgamma[c] = g; dbeta[c] = b;
    }
}

constexpr int LN_REDUCE_BATCH = 96;          // 96 x 32 B of kernel arguments: the ~52 LayerNorms of a Swin backward pass in ONE launch
struct LnReduceBatch { MvltLnReduceItem it[LN_REDUCE_BATCH]; int n; };
static_assert(sizeof(LnReduceBatch) <= 3584, "kernel argument block");
// one launch reduces the partial rows of up to 96 LayerNorms (LN_REDUCE_BATCH): blockIdx.y = item, blockIdx.x = 64-column group
__global__ __launch_bounds__(1024) void ln_param_reduce_batch_kernel(const LnReduceBatch b) {
    __shared__ float red[2][16][64];
    const MvltLnReduceItem it = b.it[blockIdx.y];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    if (blockIdx.x * 64 >= it.C) return;
    float g = 0.f, bb = 0.f;
    if (c < it.C) {
        const float* pg = it.workspace;
        const float* pb = it.workspace + (long)LN_PARTS_STRIDE * it.C;
        // four rows of each matrix in flight per step (a plain loop waits out one load latency per partial row: 39 us for
        // the 24 LayerNorms of a Swin / BERT pass of that round)
        float g1 = 0.f, g2 = 0.f, g3 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
        int i = w;
        for (; i + 48 < it.nparts; i += 64) {
            const long o = (long)i * it.C + c, s16 = 16L * it.C;
            g += pg[o]; g1 += pg[o + s16]; g2 += pg[o + 2 * s16]; g3 += pg[o + 3 * s16];
            bb += pb[o]; b1 += pb[o + s16]; b2 += pb[o + 2 * s16]; b3 += pb[o + 3 * s16];
        }
        for (; i < it.nparts; i += 16) { g += pg[(long)i * it.C + c]; bb += pb[(long)i * it.C + c]; }
        g += g1 + g2 + g3; bb += b1 + b2 + b3;
    }
    red[0][w][lane] = g; red[1][w][lane] = bb;
    __syncthreads();
    if (w == 0 && c < it.C) {
        g = 0.f; bb = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { g += red[0][i][lane]; bb += red[1][i][lane]; }
        it.dgamma[c] = g; it.dbeta[c] = bb;
    }
}

// y = LayerNorm(sum_s acc[s] + bias + residual) for a few rows (decode: 2B <= 64 rows), one wave per row; acc holds the nsplit
// k-slice slabs [nsplit][rows][C] (f32) of mvlt_gemm_skinny_accum, summed here in slice order (deterministic; nothing to zero).
// NS > 0: the slice count at compile time (all slab loads of a row are independent and go out together: one memory round trip);
// NS = 0: any count, slice by slice.
template <typename T, int NS>
__global__ __launch_bounds__(256) void ln_acc_fwd_kernel(const float* acc, int nsplit, const float* bias, const T* residual, const float* gamma,
                                                        const float* beta, float eps, int rows, int C, T* y) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* a = acc + (long)r * C;
    const long slab = (long)rows * C;
    constexpr int MAXV = 8;                           // C <= 2048
    f32x4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c = 4 * (lane + 64 * j);
        v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < C) {
            if constexpr (NS > 0) {
                f32x4 part[NS];
#pragma unroll
                for (int sl = 0; sl < NS; ++sl) part[sl] = load4f(a + sl * slab + c);
                v[j] = part[0];
#pragma unroll
                for (int sl = 1; sl < NS; ++sl) v[j] += part[sl];          // slice order: the same sum whatever NS path runs
            } else {
                v[j] = load4f(a + c);
                for (int sl = 1; sl < nsplit; ++sl) v[j] += load4f(a + sl * slab + c);
            }
            v[j] += load4f(bias + c);
            if (residual) v[j] += load4f(residual + (long)r * C + c);
            s += v[j][0] + v[j][1] + v[j][2] + v[j][3];
        }
    }
    const float mean = wave_sum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c = 4 * (lane + 64 * j);
        if (c < C) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[j][e] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / C + eps);
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c = 4 * (lane + 64 * j);
        if (c < C) {
            const f32x4 ga = load4f(gamma + c), be = load4f(beta + c);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[j][e] - mean) * rstd * ga[e] + be[e];
            store4f(y + (long)r * C + c, o);
        }
    }
}

constexpr int LN_BWD_PARTS = 512;
// MVLT_LN_BWD2=0: the round-1..5 kernel in 8-wave blocks everywhere (A/B switch; read once)
static bool ln_bwd2_on() { static const bool on = [] { const char* e = getenv("MVLT_LN_BWD2"); return !(e && e[0] == '0'); }(); return on; }
static int ln_bwd_waves(int C) {
    // 4-wave blocks for EVERY variant while the low-footprint kernel is on: the number of partial rows a launch writes
    // (mvlt_layernorm_bwd_nparts) is a function of (rows, C) alone, whichever kernel the dtype / gelu / merge flags pick
    if (ln_bwd2_on()) return 4;
    int nw = 8;                                    // per-wave partial rows: nw * 2 * C floats of LDS, keep <= 64 KB
    while (nw > 1 && (size_t)nw * 2 * C * sizeof(float) > 64 * 1024) nw >>= 1;
    return nw;
}
// Block size of the backward kernel: 8 waves and up to 512 blocks (= partial parameter-gradient rows).  Stand-alone the 16-wave /
// 256-block form is as fast; inside the step the backward runs beside the weight-gradient groups, whose workgroups hold most of
// a CU's registers and LDS, and a 1024-thread block with 49 KB of LDS waits for ALL of that to come free on one CU at once:
// 12.19 ms per step with 16 waves / 256 blocks, 11.86 with 8 / 512, 11.91 with 4 / 1024 (same box, interleaved).
static int ln_bwd_parts() { return LN_BWD_PARTS; }

// lanes per row of the instantiation dispatch() picks for a width (the ONE place that decides it: the number of partial
// parameter-gradient rows a backward launch writes follows from it)
static int ln_lpr(int C) {
    if (C == 96) return 8;
    if (C == 192) return 16;
    if (C == 384) return 32;
    return C <= 64 ? 16 : (C <= 128 ? 32 : 64);
}
static int ln_bwd_blocks(int rows, int C) {
    const int blocks = ceil_div(rows, ln_bwd_waves(C) * (64 / ln_lpr(C)));
    return blocks > ln_bwd_parts() ? ln_bwd_parts() : blocks;
}

template <typename T, int LPR, int NV>
void launch_fwd(const LnDev& d, bool merge, hipStream_t s) {
    const int rpb = 4 * (64 / LPR);
    dim3 grid(ceil_div(d.rows, rpb));
    if (merge) hipLaunchKernelGGL((ln_fwd_kernel<T, LPR, NV, true>), grid, dim3(256), 0, s, d);
    else hipLaunchKernelGGL((ln_fwd_kernel<T, LPR, NV, false>), grid, dim3(256), 0, s, d);
}
thread_local bool g_ln_parts_unsupported = false;
template <typename T, int LPR, int NV>
void launch_bwd(LnDev d, bool merge, hipStream_t s) {
    const int nw = ln_bwd_waves(d.C);
    const int blocks = ln_bwd_blocks(d.rows, d.C);       // (LPR == ln_lpr(C): dispatch() below)
    d.nparts = blocks;
    if constexpr (sizeof(T) == 2 && NV <= 4) {
        // the 150 launches of a step; LayerNorm + GELU, the patch-merging gather, f32 and widths that do not fill the lanes
        // exactly keep the general kernel (in 4-wave blocks too)
        if (ln_bwd2_on() && !merge && !d.gelu && d.C == 4 * LPR * NV && (long)d.rows * d.C < (1L << 30) - 65536) {
            static const bool pf = [] { const char* e = getenv("MVLT_LN_BWD2_PF"); return e && e[0] == '1'; }();          // two row groups in flight per wave (A/B switch)
            const size_t sh2 = 3 * (size_t)d.C * sizeof(float);          // reduction rows [2][C] + gamma [C]
            if (d.dy_parts > 1) hipLaunchKernelGGL((ln_bwd2_kernel<LPR, NV, false, true>), dim3(blocks), dim3(256), sh2, s, d);
            else if (pf && NV <= 3) hipLaunchKernelGGL((ln_bwd2_kernel<LPR, NV, true>), dim3(blocks), dim3(256), sh2, s, d);
            else hipLaunchKernelGGL((ln_bwd2_kernel<LPR, NV, false>), dim3(blocks), dim3(256), sh2, s, d);
            return;
        }
    }
    if (d.dy_parts > 1) { g_ln_parts_unsupported = true; return; }
    const size_t sh = 2 * (size_t)nw * d.C * sizeof(float);
    if (merge) hipLaunchKernelGGL((ln_bwd_kernel<T, LPR, NV, true>), dim3(blocks), dim3(64 * nw), sh, s, d);
    else hipLaunchKernelGGL((ln_bwd_kernel<T, LPR, NV, false>), dim3(blocks), dim3(64 * nw), sh, s, d);
}

template <typename T, bool BWD>
int dispatch(const LnDev& d, bool merge, hipStream_t s) {
    const int C = d.C;
#define LN_CASE(LPR, NV) do { if (BWD) launch_bwd<T, LPR, NV>(d, merge, s); else launch_fwd<T, LPR, NV>(d, merge, s); } while (0)
    // widths that are 12 x a power of two (the Swin-S stages) fit three 4-element chunks per lane exactly: more rows per wave
    // (the per-row reduction chain, not the bytes, is what these kernels wait for) and no idle lanes
    // (keep in step with ln_lpr above)
    if (C == 96) LN_CASE(8, 3);
    else if (C == 192) LN_CASE(16, 3);
    else if (C == 384) LN_CASE(32, 3);
    else if (C <= 64) LN_CASE(16, 1);
    else if (C <= 128) LN_CASE(32, 1);
    else if (C <= 256) LN_CASE(64, 1);
    else if (C <= 512) LN_CASE(64, 2);
    else if (C <= 768) LN_CASE(64, 3);
    else if (C <= 1024) LN_CASE(64, 4);
    else if (C <= 1536) LN_CASE(64, 6);
    else if (C <= 2048) LN_CASE(64, 8);
    else return MVLT_ERR_UNSUPPORTED;
#undef LN_CASE
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}

}  // namespace

extern "C" int mvlt_layernorm_bwd_workspace_rows(void) { return LN_BWD_PARTS; }

extern "C" int mvlt_layernorm_acc_fwd(int dtype, const float* acc, int nsplit, const float* bias, const void* residual, const float* gamma,
                                      const float* beta, float eps, int rows, int C, void* y, void* stream) {
    MVLT_CHECK(acc && nsplit >= 1 && nsplit <= 64 && bias && gamma && beta && y && rows > 0 && C > 0 && C % 4 == 0 && C <= 2048, MVLT_ERR_ARG);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid(ceil_div(rows, 4));
#define LN_ACC_GO(T_, NS_) hipLaunchKernelGGL((ln_acc_fwd_kernel<T_, NS_>), grid, dim3(256), 0, s, acc, nsplit, bias, (const T_*)residual, gamma, beta, eps, rows, C, (T_*)y)
#define LN_ACC_BY_NS(T_) do { if (nsplit == 1) LN_ACC_GO(T_, 1); else if (nsplit == 2) LN_ACC_GO(T_, 2); else if (nsplit == 4) LN_ACC_GO(T_, 4); else LN_ACC_GO(T_, 0); } while (0)
    if (dtype == MVLT_BF16) LN_ACC_BY_NS(bf16_t);
    else if (dtype == MVLT_F32) LN_ACC_BY_NS(float);
#undef LN_ACC_BY_NS
#undef LN_ACC_GO
    else return MVLT_ERR_UNSUPPORTED;
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}

extern "C" int mvlt_layernorm_fwd(const MvltLayerNorm* p, void* stream) {
    MVLT_CHECK(p && p->x && p->y && p->gamma && p->beta, MVLT_ERR_ARG);
    MVLT_CHECK(p->rows > 0 && p->C > 0 && p->C % 4 == 0, MVLT_ERR_ARG);
    const bool merge = p->merge_H != 0;
    if (merge) MVLT_CHECK(p->merge_H % 2 == 0 && p->merge_W % 2 == 0 && p->C % 16 == 0 &&
                          p->rows % ((p->merge_H / 2) * (p->merge_W / 2)) == 0, MVLT_ERR_ARG);
    LnDev d{};
    d.rows = p->rows; d.C = p->C; d.eps = p->eps; d.x = p->x; d.gamma = p->gamma; d.beta = p->beta;
    d.y = p->y; d.y_pre = p->y_pre; d.mean = p->mean; d.rstd = p->rstd; d.rowmap = p->out_rowmap;
    d.mH = p->merge_H; d.mW = p->merge_W; d.gelu = p->gelu; d.rows_dev = p->rows_dev;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (p->dtype == MVLT_F32) return dispatch<float, false>(d, merge, s);
    if (p->dtype == MVLT_BF16) return dispatch<bf16_t, false>(d, merge, s);
    return MVLT_ERR_UNSUPPORTED;
}

extern "C" int mvlt_layernorm_bwd(const MvltLayerNormBwd* p, void* stream) {
    MVLT_CHECK(p && p->dy && p->x && p->mean && p->rstd && p->gamma && p->dx, MVLT_ERR_ARG);
    MVLT_CHECK(p->dgamma && p->dbeta && p->workspace, MVLT_ERR_ARG);
    MVLT_CHECK(p->rows > 0 && p->C > 0 && p->C % 4 == 0, MVLT_ERR_ARG);
    if (p->gelu) MVLT_CHECK(p->y_pre, MVLT_ERR_ARG);
    const bool merge = p->merge_H != 0;
    if (merge) MVLT_CHECK(p->merge_H % 2 == 0 && p->merge_W % 2 == 0 && p->C % 16 == 0 && !p->dres && !p->dz, MVLT_ERR_ARG);
    MVLT_CHECK(p->dz_dropout_p >= 0.f && p->dz_dropout_p < 1.f, MVLT_ERR_ARG);
    LnDev d{};
    d.rows = p->rows; d.C = p->C; d.x = p->x; d.gamma = p->gamma; d.mean = const_cast<float*>(p->mean);
    d.rstd = const_cast<float*>(p->rstd); d.rowmap = p->dy_rowmap; d.mH = p->merge_H; d.mW = p->merge_W;
    d.gelu = p->gelu; d.y_pre = const_cast<void*>(p->y_pre); d.dy = p->dy; d.dres = p->dres; d.dx = p->dx;
    d.part_g = p->workspace; d.part_b = p->workspace + (size_t)LN_BWD_PARTS * p->C;
    d.dz = p->dz; d.zmap = p->dz_rowmap; d.zscale = p->dz_rowscale; d.zrps = p->dz_rows_per_scale > 0 ? p->dz_rows_per_scale : 1;
    d.zthresh = (uint32_t)((double)p->dz_dropout_p * 4294967296.0); d.zdscale = 1.0f / (1.0f - p->dz_dropout_p);
    d.seed = p->seed; d.tag = p->tag; d.rows_dev = p->rows_dev;
    if (p->dy_parts > 1) {
        // the low-footprint bf16 kernel only (what mvlt_swin_wmsa2_bwd feeds): 2 .. 4 parts, no GELU, no merge
        MVLT_CHECK(p->dy_parts <= 4 && p->dy_part_stride >= (int64_t)p->rows * p->C, MVLT_ERR_ARG);
        if (p->dtype != MVLT_BF16 || merge || p->gelu || !ln_bwd2_on() || (double)p->dy_parts * p->dy_part_stride * 2 >= 2147483648.0) return MVLT_ERR_UNSUPPORTED;
        d.dy_parts = p->dy_parts; d.dy_part_bytes = (long)p->dy_part_stride * 2;
    }
    g_ln_parts_unsupported = false;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int rc;
    if (p->dtype == MVLT_F32) rc = dispatch<float, true>(d, merge, s);
    else if (p->dtype == MVLT_BF16) rc = dispatch<bf16_t, true>(d, merge, s);
    else return MVLT_ERR_UNSUPPORTED;
    if (rc != MVLT_OK) return rc;
    if (g_ln_parts_unsupported) return MVLT_ERR_UNSUPPORTED;         // (a width the low-footprint kernel does not take: nothing was launched)
    // the number of partial rows written == number of blocks launched above
    const int blocks = ln_bwd_blocks(p->rows, p->C);
    if (p->defer_param_reduce) return MVLT_OK;       // caller batches it with mvlt_layernorm_param_reduce_batch
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3(ceil_div(p->C, 64)), dim3(1024), 0, s, d.part_g, d.part_b,
                       blocks, p->C, p->dgamma, p->dbeta, p->accumulate);
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}

extern "C" int mvlt_layernorm_bwd_nparts(int rows, int C) {
    return ln_bwd_blocks(rows, C);
}

extern "C" int mvlt_layernorm_param_reduce_batch(const MvltLnReduceItem* items, int n, void* stream) {
    MVLT_CHECK(items && n > 0, MVLT_ERR_ARG);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    for (int i0 = 0; i0 < n; i0 += LN_REDUCE_BATCH) {
        LnReduceBatch b;
        b.n = n - i0 < LN_REDUCE_BATCH ? n - i0 : LN_REDUCE_BATCH;
        int maxC = 0;
        for (int i = 0; i < b.n; ++i) {
            b.it[i] = items[i0 + i];
            MVLT_CHECK(b.it[i].workspace && b.it[i].dgamma && b.it[i].dbeta && b.it[i].nparts > 0 && b.it[i].C > 0, MVLT_ERR_ARG);
            if (b.it[i].C > maxC) maxC = b.it[i].C;
        }
        hipLaunchKernelGGL(ln_param_reduce_batch_kernel, dim3(ceil_div(maxC, 64), b.n), dim3(1024), 0, s, b);
        MVLT_LAUNCH_CHECK();
    }
    return MVLT_OK;
}
