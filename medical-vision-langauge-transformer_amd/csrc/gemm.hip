// MFMA GEMM for gfx950 with fused epilogues (see include/mvlt_hip.h, MvltGemm).
//
// Structure: 256 threads = 4 waves (2 x 2), workgroup tile BM x BN, k-tile of
// 128 bytes per row (64 bf16 / 32 f32 = two MFMA k-blocks), register-staged
// double-buffered LDS (global loads for tile t+1 are issued before the MFMAs
// of tile t and written to the other LDS buffer after them: one barrier per
// k-tile).  k-contiguous operands use an XOR-swizzled [row][128 B] image read
// with ds_read_b128 (conflict-free); k-strided operands (dgrad weights, both
// wgrad operands) are staged as they lie in memory ([k][row], padded rows) and
// transposed for free by ds_read_b64_tr_b16 (bf16) when the fragment is read.
// The MFMA is issued as (B-fragment, A-fragment) so each lane owns 4
// consecutive n of one output row -> 8/16-byte epilogue loads and stores.
#include "common.h"
#include "gemm_dev.h"
#include <cstdio>
#include <cstdlib>

// gemm8.hip: 8-wave ping-pong kernel; takes the filled kernel argument block, returns 1 if it launched
extern "C" __attribute__((visibility("hidden"))) int mvlt_gemm8_try(const void* dev_blocks, int n, int a_kmajor, int b_kmajor, int big_only,
                                                                    float* const* colsum, void* ws, size_t ws_bytes, void* stream);
extern "C" __attribute__((visibility("hidden"))) size_t mvlt_gemm8_group_workspace(const void* dev_blocks, int n);
// row-streaming kernel for the HBM-bound Swin stage-0 / 1 products (rowstream.hip): 1 = taken, 0 = not eligible, -1 = error
extern "C" __attribute__((visibility("hidden"))) int mvlt_rowstream_try(const void* dev_block, int b_kmajor, void* stream);
// MVLT_G8: unset / 2 = automatic (single products that fill the chip with 256 x 256 tiles: MLM decoder, large batches),
// 0 = never, 1 = wherever it is eligible, weight-gradient groups included (experiments: slower than the 4-wave kernels on
// the B=32 step's mid-size products)
static int g8_mode() { const char* e = getenv("MVLT_G8"); return e ? atoi(e) : 2; }

namespace {

// k-major bf16 tiles of 64 / 96 / 128 rows are stored (96: in 128-wide rows) with their 32-byte column chunks XOR-swizzled by a function
// of k: a ds_read_b64_tr_b16 group of 32 lanes reads 8 k-rows {k0..k0+3, k0+8..k0+11} x 32 bytes, which padding alone
// cannot spread over the 64 banks (rows k and k+8 alias for every pad that keeps 32-byte chunks aligned: 2-way conflicts,
// 32 % of the LDS cycles of the weight-gradient kernels, profiles/r2_dominant_kernel_pmc.txt).
template <int R> MVLT_DEV int kswz(int k) {
    return R >= 96 ? ((k & 3) | ((k >> 1) & 4)) : (((k >> 1) & 1) | ((k >> 2) & 2));
}
template <typename T, int R, bool KMAJOR> struct TileGeom {
    static constexpr int E = TypeInfo<T>::E;
    static constexpr int BKE = 128 / (int)sizeof(T);          // k elements per tile
    static constexpr bool SWZ = KMAJOR && sizeof(T) == 2 && (R == 64 || R == 96 || R == 128);
    static constexpr int PAD = KMAJOR ? (SWZ ? (R == 96 ? 32 : 0) : (sizeof(T) == 2 ? 16 : 4)) : 0;   // 96: six chunks swizzled inside eight
    static constexpr int LD = KMAJOR ? (R + PAD) : BKE;        // elements per LDS row
    static constexpr int ELEMS = KMAJOR ? BKE * LD : R * BKE;
    static constexpr int CHUNKS = R * 8;                        // 16-byte chunks per tile
    static constexpr int PER_THREAD = CHUNKS / 256;
    static constexpr int CPR = KMAJOR ? (R / E) : 8;            // chunks per LDS row
};

template <typename T, int R, bool KMAJOR>
MVLT_DEV void tile_load(typename TypeInfo<T>::Vec* __restrict__ regs, const T* base,
                        long ld, int row0, int row_lim, int k0, int k_lim, bool vec_ok) {
    using G = TileGeom<T, R, KMAJOR>;
#pragma unroll
    for (int i = 0; i < G::PER_THREAD; ++i) {
        const int idx = threadIdx.x + 256 * i;
        const int lr = idx / G::CPR, ch = idx % G::CPR;
        if (KMAJOR) regs[i] = load_chunk<T>(base, ld, k0 + lr, row0 + ch * G::E, k_lim, row_lim, vec_ok);
        else        regs[i] = load_chunk<T>(base, ld, row0 + lr, k0 + ch * G::E, row_lim, k_lim, vec_ok);
    }
}

// Fast path of the k-loop: per-thread source pointers are computed ONCE (rows /
// columns beyond the matrix edge are clamped to the last valid one: they only
// feed output rows/columns that are never stored), then every full k-tile is
// PER_THREAD unpredicated 16-byte loads and a pointer bump.
template <typename T, int R, bool KMAJOR> struct FastLoader {
    using G = TileGeom<T, R, KMAJOR>;
    using Vec = typename TypeInfo<T>::Vec;
    const T* ptr[G::PER_THREAD];
    long step;
    MVLT_DEV void init(const T* base, long ld, int row0, int row_lim, int k0) {
#pragma unroll
        for (int i = 0; i < G::PER_THREAD; ++i) {
            const int idx = threadIdx.x + 256 * i;
            const int lr = idx / G::CPR, ch = idx % G::CPR;
            if (KMAJOR) {
                int col = row0 + ch * G::E;
                col = min(col, max(row_lim - G::E, 0));
                ptr[i] = base + (long)(k0 + lr) * ld + col;
            } else {
                const int row = min(row0 + lr, row_lim - 1);
                ptr[i] = base + (long)row * ld + k0 + ch * G::E;
            }
        }
        step = KMAJOR ? (long)G::BKE * ld : (long)G::BKE;
    }
    MVLT_DEV void load(Vec* __restrict__ regs) {
#pragma unroll
        for (int i = 0; i < G::PER_THREAD; ++i) { regs[i] = *reinterpret_cast<const Vec*>(ptr[i]); ptr[i] += step; }
    }
};

template <typename T, int R, bool KMAJOR>
MVLT_DEV void tile_store(const typename TypeInfo<T>::Vec* __restrict__ regs, T* lds) {
    using G = TileGeom<T, R, KMAJOR>;
    using Vec = typename TypeInfo<T>::Vec;
#pragma unroll
    for (int i = 0; i < G::PER_THREAD; ++i) {
        const int idx = threadIdx.x + 256 * i;
        const int lr = idx / G::CPR, ch = idx % G::CPR;
        if (KMAJOR && G::SWZ) *reinterpret_cast<Vec*>(lds + lr * G::LD + ((((ch >> 1) ^ kswz<R>(lr)) << 4) | ((ch & 1) << 3))) = regs[i];
        else if (KMAJOR) *reinterpret_cast<Vec*>(lds + lr * G::LD + ch * G::E) = regs[i];
        else        *reinterpret_cast<Vec*>(lds + lr * G::BKE + ((ch ^ (lr & 7)) * G::E)) = regs[i];
    }
}

// fragment for rows [row0,row0+16) and k-block kb (0/1) of the tile
template <typename T, int R, bool KMAJOR>
MVLT_DEV typename Mma<T>::Frag tile_frag(const T* lds, int row0, int kb) {
    using G = TileGeom<T, R, KMAJOR>;
    if constexpr (KMAJOR && G::SWZ) {
        const int l = threadIdx.x & 63;
        const int g = l >> 4, i = l & 15, q = i >> 2, pp = i & 3;
        const int k = kb * 32 + 8 * g + q, c = row0 >> 4;
        const bf16_t* p0 = lds + k * G::LD + ((c ^ kswz<R>(k)) << 4) + 4 * pp;
        const bf16_t* p1 = lds + (k + 4) * G::LD + ((c ^ kswz<R>(k + 4)) << 4) + 4 * pp;
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p0);
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p1);
        bf16x8 r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return r;
    } else if constexpr (KMAJOR) {
        return frag_kmajor(lds, G::LD, row0, kb * Mma<T>::KB);
    } else {
    const int l = threadIdx.x & 63;
    const int row = row0 + (l & 15);
    const int ch = (kb * 4 + (l >> 4)) ^ (row & 7);
    return *reinterpret_cast<const typename Mma<T>::Frag*>(lds + row * G::BKE + ch * G::E);
    }
}

// one output tile (bx, by) of one k-split bz
template <typename T, int BM, int BN, bool AK, bool BK_, bool PF2, int DEEP = 0, bool WIDE = false>
MVLT_DEV void gemm_body(const GemmDev& p_in, const int bx, const int by, const int bz, T* sA, T* sB) {
    using GA = TileGeom<T, BM, AK>;
    using GB = TileGeom<T, BN, BK_>;
    using Vec = typename TypeInfo<T>::Vec;
    constexpr int FM = BM / 32, FN = BN / 32;
    const GemmDev& p = p_in;                     // (already the effective problem: see gemm_kernel / gemm_group_kernel)
    const int m0 = by * BM, n0 = bx * BN;
    if (m0 >= p.M) return;                       // (uniform per workgroup; only with m_dev)
    const int ks = bz * p.k_per_split;
    const int ke = min(p.K, ks + p.k_per_split);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const T* A = reinterpret_cast<const T*>(p.A);
    const T* B = reinterpret_cast<const T*>(p.B);

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Single LDS stage (so 3-4 workgroups fit per CU and hide each other's HBM latency);
    // the next tile's global loads are in flight in registers while this tile is multiplied.
    Vec ra[GA::PER_THREAD], rb[GB::PER_THREAD];
    const int nkt = (ke - ks + GA::BKE - 1) / GA::BKE;
    // number of k-tiles that lie fully inside [ks, ke) and can use unpredicated vector loads
    const int nfast = (p.a_vec && p.b_vec && (AK || p.M >= 1) ) ? (ke - ks) / GA::BKE : 0;
    FastLoader<T, BM, AK> la;
    FastLoader<T, BN, BK_> lb;
    la.init(A, p.lda, m0, p.M, ks);
    lb.init(B, p.ldb, n0, p.N, ks);
    // fused bias gradient: workgroups of the first output-column tile also sum their A tile over k.  The sums are
    // taken from the staging REGISTERS (a thread's chunks of a k-major tile all lie in one 16-byte column group:
    // 256 % CPR == 0), a few adds per k-tile spread over all 256 threads, and meet in LDS once at the end; summing
    // the LDS image instead (64 dependent ds_reads per k-tile in 64-128 threads while the other waves wait at the
    // barrier) cost 35-50 % of the weight-gradient launches (profiles/r2_wgrad_colsum.txt).
    constexpr int EA = GA::E;
    static_assert(!AK || (256 % GA::CPR == 0 && (size_t)GA::ELEMS * sizeof(T) >= 256 * EA * sizeof(float)), "colsum layout");
    float cs[EA];
#pragma unroll
    for (int e = 0; e < EA; ++e) cs[e] = 0.f;
    const bool do_colsum = AK && p.a_colsum != nullptr && bx == 0;
    auto colsum_regs = [&](const Vec* r) {
        if (AK && do_colsum) {
#pragma unroll
            for (int i = 0; i < GA::PER_THREAD; ++i)
#pragma unroll
                for (int e = 0; e < EA; ++e) cs[e] += to_f(r[i][e]);
        }
    };
    auto compute_tile = [&]() {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            typename Mma<T>::Frag fa[FM], fb[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) fa[i] = tile_frag<T, BM, AK>(sA, wm * (BM / 2) + i * 16, kb);
#pragma unroll
            for (int j = 0; j < FN; ++j) fb[j] = tile_frag<T, BN, BK_>(sB, wn * (BN / 2) + j * 16, kb);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) Mma<T>::mma(acc[i][j], fb[j], fa[i]);
        }
    };
    // ---- hot loop: full k-tiles only, no predication, nothing but loads / ds_write / ds_read / MFMA
    if constexpr (DEEP == 2) {
        // weight gradients (1-2 workgroups per CU, thousands of k rows): two register sets AND two LDS stages, one
        // barrier per k-tile.  Tile t+1 (loaded a whole iteration ago) is written to the other LDS stage at the top of
        // iteration t, so its ds_writes, the global loads of tile t+2 and the ds_reads + MFMAs of tile t all overlap;
        // with one stage every k-tile was a chain load-wait -> ds_write -> barrier -> ds_read -> MFMA -> barrier.
        Vec qa[2][GA::PER_THREAD], qb[2][GB::PER_THREAD];
        T* const sA1 = sA + GA::ELEMS;
        T* const sB1 = sB + GB::ELEMS;
        auto compute_stage = [&](const T* a_, const T* b_) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                typename Mma<T>::Frag fa[FM], fb[FN];
#pragma unroll
                for (int i = 0; i < FM; ++i) fa[i] = tile_frag<T, BM, AK>(a_, wm * (BM / 2) + i * 16, kb);
#pragma unroll
                for (int j = 0; j < FN; ++j) fb[j] = tile_frag<T, BN, BK_>(b_, wn * (BN / 2) + j * 16, kb);
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) Mma<T>::mma(acc[i][j], fb[j], fa[i]);
            }
        };
        if (nfast > 0) {
            la.load(qa[0]); lb.load(qb[0]);
            if (nfast > 1) { la.load(qa[1]); lb.load(qb[1]); }
            tile_store<T, BM, AK>(qa[0], sA);
            tile_store<T, BN, BK_>(qb[0], sB);
            colsum_regs(qa[0]);
            __syncthreads();
            for (int kt = 0; kt < nfast; kt += 2) {
                // even tile kt lives in stage 0, registers set 0 is free again
                if (kt + 1 < nfast) { tile_store<T, BM, AK>(qa[1], sA1); tile_store<T, BN, BK_>(qb[1], sB1); colsum_regs(qa[1]); }
                if (kt + 2 < nfast) { la.load(qa[0]); lb.load(qb[0]); }
                compute_stage(sA, sB);
                __syncthreads();
                if (kt + 1 < nfast) {
                    if (kt + 2 < nfast) { tile_store<T, BM, AK>(qa[0], sA); tile_store<T, BN, BK_>(qb[0], sB); colsum_regs(qa[0]); }
                    if (kt + 3 < nfast) { la.load(qa[1]); lb.load(qb[1]); }
                    compute_stage(sA1, sB1);
                    __syncthreads();
                }
            }
        }
    } else if (!PF2) {
        if (nfast > 0) { la.load(ra); lb.load(rb); }
        for (int kt = 0; kt < nfast; ++kt) {
            tile_store<T, BM, AK>(ra, sA);
            tile_store<T, BN, BK_>(rb, sB);
            colsum_regs(ra);
            __syncthreads();
            if (kt + 1 < nfast) { la.load(ra); lb.load(rb); }
            compute_tile();
            __syncthreads();
        }
    } else {
        // long reductions: prefetch distance 2 (two register sets) -- the loads of tile t+2 are in
        // flight while tile t is multiplied (measured 8-17 % faster for K >= 1536)
        Vec ra1[GA::PER_THREAD], rb1[GB::PER_THREAD];
        if (nfast > 0) { la.load(ra); lb.load(rb); }
        if (nfast > 1) { la.load(ra1); lb.load(rb1); }
        for (int kt = 0; kt < nfast; kt += 2) {
            tile_store<T, BM, AK>(ra, sA);
            tile_store<T, BN, BK_>(rb, sB);
            colsum_regs(ra);
            __syncthreads();
            if (kt + 2 < nfast) { la.load(ra); lb.load(rb); }
            compute_tile();
            __syncthreads();
            if (kt + 1 < nfast) {
                tile_store<T, BM, AK>(ra1, sA);
                tile_store<T, BN, BK_>(rb1, sB);
                colsum_regs(ra1);
                __syncthreads();
                if (kt + 3 < nfast) { la.load(ra1); lb.load(rb1); }
                compute_tile();
                __syncthreads();
            }
        }
    }
    // ---- K-tail (at most one tile when the operands are vector-aligned), generic predicated loads
    for (int kt = nfast; kt < nkt; ++kt) {
        const int k0 = ks + kt * GA::BKE;
        tile_load<T, BM, AK>(ra, A, p.lda, m0, p.M, k0, ke, p.a_vec);
        tile_load<T, BN, BK_>(rb, B, p.ldb, n0, p.N, k0, ke, p.b_vec);
        tile_store<T, BM, AK>(ra, sA);
        tile_store<T, BN, BK_>(rb, sB);
        colsum_regs(ra);
        __syncthreads();
        compute_tile();
        __syncthreads();
    }
    if (AK && do_colsum) {
        float* red = reinterpret_cast<float*>(sA);           // free: the k-loop ended on a barrier
#pragma unroll
        for (int e = 0; e < EA; ++e) red[threadIdx.x * EA + e] = cs[e];
        __syncthreads();
        float csum = 0.f;
        if (threadIdx.x < BM) {
            const int ch = threadIdx.x / EA, e = threadIdx.x % EA;
#pragma unroll 4
            for (int g = 0; g < 256 / GA::CPR; ++g) csum += red[(g * GA::CPR + ch) * EA + e];
        }
        const int m = m0 + threadIdx.x;
        if (threadIdx.x < BM && m < p.M) {
            if (p.atomic_out) atomicAdd(&p.a_colsum[m], csum);
            else if (p.split_k > 1) p.ws_colsum[(long)bz * p.M + m] = csum;
            else p.a_colsum[m] = (p.epi & MVLT_EPI_ACCUM) ? p.a_colsum[m] + csum : csum;
        }
        __syncthreads();                                      // (persistent callers reuse sA)
    }

    // acc[i][j][r] <-> n = nb + 4*(lane>>4) + r, m = mb + (lane & 15)
    if (!p.atomic_out && p.split_k <= 1 && p.epi_vec && (p.N & 3) == 0) {
        if constexpr (WIDE && sizeof(T) == 2 && FN % 2 == 0) tile_epilogue_wide<T, FM, FN>(p, m0 + wm * (BM / 2), n0 + wn * (BN / 2), acc);
        else tile_epilogue<T, FM, FN>(p, m0 + wm * (BM / 2), n0 + wn * (BN / 2), acc);          // loads hoisted out of the store sequence
        return;
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + wm * (BM / 2) + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = n0 + wn * (BN / 2) + j * 16 + 4 * (lane >> 4);
            if (p.atomic_out) {
                // k-slices of one output tile meet in the f32 output itself (zeroed by the launcher): no slabs, no
                // reduce launch; the order of the 2..8 additions per element is not fixed (last-bit differences)
                if (m < p.M) {
                    float* c = reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n;
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < p.N) atomicAdd(c + r, acc[i][j][r]);
                }
            } else if (p.split_k > 1) {
                if (m < p.M && n < p.N) {
                    float* w = p.ws + ((long)bz * p.M + m) * p.N + n;
                    if ((p.N & 3) == 0) store4f(w, acc[i][j]);
                    else for (int r = 0; r < 4; ++r) if (n + r < p.N) w[r] = acc[i][j][r];
                }
            } else {
                epilogue4<T>(p, m, n, acc[i][j]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Both operands k-contiguous (every forward nn.Linear), bf16, K a multiple of 64: the tiles go global -> LDS by
// `global_load_lds_dwordx4` (LDS-DMA: no staging registers, no ds_write pass -- the register-staged loop spends
// about as long in ds_write_b128 as in MFMA), two LDS buffers, ONE barrier per k-tile: the DMA of tile t+1 is in
// flight while tile t is multiplied.  An LDS-DMA instruction writes wave-uniform base + lane * 16 bytes, so the
// XOR swizzle of the [row][128 B] image is applied to the per-lane SOURCE address (chunk' = chunk ^ (row & 7), the
// same involution tile_frag applies on the read side).
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

template <int R>
MVLT_DEV void glds_fill(const bf16_t* const (&src)[R / 32], bf16_t* lds_tile, int wave, long koff) {
#pragma unroll
    for (int j = 0; j < R / 32; ++j) {
        bf16_t* dst = lds_tile + (wave * (R / 32) + j) * 8 * 64;          // 8 rows x 64 elements = 1 KB per instruction
        __builtin_amdgcn_global_load_lds((glb_void_t*)(src[j] + koff), (lds_void_t*)dst, 16, 0, 0);
    }
}

// BKM: the B operand is k-major (dgrad: dx = dy W reads the [N_out, K_in] weight along its rows).  Its tile image is
// [64 k][BN] with the 32-byte units XOR-swizzled by kswz<BN>(k) (the layout tile_frag<T, BN, true> transposes out of with
// ds_read_b64_tr_b16); one LDS-DMA instruction covers 1 KB = 8 (BN = 64) / 4 (BN = 128) k-rows, the swizzle again on the
// source side.  With transposing reads in the loop the DMA is issued through inline asm (glds16_asm, gemm_dev.h).
template <int N> MVLT_DEV void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int BM, int BN, bool BKM, bool WIDE, int NST = 2>
__global__ __launch_bounds__(256, 2) void gemm_glds_kernel(const GemmDev p_in) {
    using T = bf16_t;
    constexpr int BKE = 64, FM = BM / 32, FN = BN / 32;
    static_assert(!BKM || BN == 64 || BN == 128, "k-major B tiles: 64 or 128 columns");
    static_assert(NST >= 2 && NST <= 5 && (NST - 2) * (BM / 32 + BN / 32) < 64, "two to five LDS stages (counted vmcnt: 6 bits)");
    // NST stages; beyond the 64 KB a static array may have (160 x 128 tiles: 72 KB) the launcher passes dynamic LDS
    constexpr bool DYN = NST * (BM + BN) * BKE * sizeof(T) > 65536;
    extern __shared__ __attribute__((aligned(16))) char glds_dyn[];
    __shared__ __attribute__((aligned(16))) T glds_static[DYN ? 8 : NST * (BM + BN) * BKE];
    T* const smem = DYN ? reinterpret_cast<T*>(glds_dyn) : glds_static;
    const GemmDev p = effective<false>(p_in);
    const unsigned pfv = prefetch_lines(p_in);
    const int gx = gridDim.x;
    const int gy = min((int)gridDim.y, (p.M + BM - 1) / BM);
    const int orig = blockIdx.y * gx + blockIdx.x;
    if (orig >= gx * gy) return;
    const int t = xcd_remap(orig, gx * gy);
    int by, bx;
    tile_coords(t, gx, gy, p.xcs, by, bx);
    const int bz = blockIdx.z;
    const int m0 = by * BM, n0 = bx * BN;
    const int ks = bz * p.k_per_split;
    const int ke = min(p.K, ks + p.k_per_split);
    const int nkt = (ke - ks) / BKE;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wm = wave >> 1, wn = wave & 1;
    const T* A = reinterpret_cast<const T*>(p.A);
    const T* B = reinterpret_cast<const T*>(p.B);
    constexpr int STAGE = (BM + BN) * BKE;            // elements per LDS stage: A tile, then B tile
    // per-lane source pointers: instruction j of this wave covers tile rows 8 (wave R/32 + j) .. +8; lane = (row_in, chunk')
    const int rin = lane >> 3, chs = (lane & 7) ^ rin;
    const T* srcA[BM / 32];
    const T* srcB[BN / 32];
#pragma unroll
    for (int j = 0; j < BM / 32; ++j) {
        const int row = min(m0 + (wave * (BM / 32) + j) * 8 + rin, p.M - 1);
        srcA[j] = A + (long)row * p.lda + ks + chs * 8;
    }
#pragma unroll
    for (int j = 0; j < BN / 32; ++j) {
        if constexpr (BKM) {
            constexpr int KPI = 512 / BN;                                    // k-rows per instruction (1 KB / (BN * 2 B))
            constexpr int XM = BN / 8 - 1;                                   // 16-byte positions per k-row - 1
            const int krow = (wave * (BN / 32) + j) * KPI + lane / (BN / 8), x = lane & XM;
            const int ch = (((x >> 1) ^ kswz<BN>(krow)) << 1) | (x & 1);
            srcB[j] = B + (long)(ks + krow) * p.ldb + min(n0 + ch * 8, max(p.N - 8, 0));
        } else {
            const int row = min(n0 + (wave * (BN / 32) + j) * 8 + rin, p.N - 1);
            srcB[j] = B + (long)row * p.ldb + ks + chs * 8;
        }
    }
    const long kstep_b = BKM ? (long)BKE * p.ldb : (long)BKE;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) T*)smem;
    auto fill = [&](int stage, int kt) {
        const unsigned sa = lds0 + (unsigned)(stage * STAGE) * 2u, sb = sa + (unsigned)(BM * BKE) * 2u;
#pragma unroll
        for (int j = 0; j < BM / 32; ++j) glds16_asm(srcA[j] + (long)kt * BKE, sa + (wave * (BM / 32) + j) * 1024);
#pragma unroll
        for (int j = 0; j < BN / 32; ++j) glds16_asm(srcB[j] + kt * kstep_b, sb + (wave * (BN / 32) + j) * 1024);
    };
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < NST - 1; ++st) if (st < nkt) fill(st, st);
    constexpr int PER = BM / 32 + BN / 32;               // LDS-DMA requests per wave and K-tile
    for (int kt = 0; kt < nkt; ++kt) {
        // this wave's share of tile kt has landed (NST stages: the requests of the next NST - 2 tiles may still be in flight --
        // vmcnt completes in order, so a COUNTED wait leaves exactly those outstanding; fewer near the end of the reduction)
        const int ahead = min(NST - 2, nkt - 1 - kt);
        if (NST >= 5 && ahead >= 3) wait_vmcnt<(NST >= 5 ? 3 : 0) * PER>();
        else if (NST >= 4 && ahead == 2) wait_vmcnt<(NST >= 4 ? 2 : 0) * PER>();
        else if (NST >= 3 && ahead == 1) wait_vmcnt<(NST >= 3 ? 1 : 0) * PER>();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                       // everybody's has; everybody is done reading tile kt-1
        if (kt + NST - 1 < nkt) fill((kt + NST - 1) % NST, kt + NST - 1);
        const T* a = smem + (kt % NST) * STAGE;
        const T* b = a + BM * BKE;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            typename Mma<T>::Frag fa[FM], fb[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) fa[i] = tile_frag<T, BM, false>(a, wm * (BM / 2) + i * 16, kb);
#pragma unroll
            for (int j = 0; j < FN; ++j) fb[j] = tile_frag<T, BN, BKM>(b, wn * (BN / 2) + j * 16, kb);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) Mma<T>::mma(acc[i][j], fb[j], fa[i]);
        }
    }
    if (p.split_k <= 1 && p.epi_vec && (p.N & 3) == 0) {
        if constexpr (WIDE && sizeof(T) == 2 && FN % 2 == 0) tile_epilogue_wide<T, FM, FN>(p, m0 + wm * (BM / 2), n0 + wn * (BN / 2), acc);
        else tile_epilogue<T, FM, FN>(p, m0 + wm * (BM / 2), n0 + wn * (BN / 2), acc);          // loads hoisted out of the store sequence
    } else {
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + wm * (BM / 2) + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = n0 + wn * (BN / 2) + j * 16 + 4 * (lane >> 4);
            if (p.split_k > 1) {
                if (m < p.M && n < p.N) {
                    float* w = p.ws + ((long)bz * p.M + m) * p.N + n;
                    if ((p.N & 3) == 0) store4f(w, acc[i][j]);
                    else for (int r = 0; r < 4; ++r) if (n + r < p.N) w[r] = acc[i][j][r];
                }
            } else {
                epilogue4<T>(p, m, n, acc[i][j]);
            }
        }
    }
    }
    asm volatile("" :: "v"(pfv));
}

template <typename T, int BM, int BN, bool AK, bool BK_, bool PF2, bool WIDE = false>
__global__ __launch_bounds__(256, 3) void gemm_kernel(const GemmDev p) {
    __shared__ __attribute__((aligned(16))) T sA[TileGeom<T, BM, AK>::ELEMS];
    __shared__ __attribute__((aligned(16))) T sB[TileGeom<T, BN, BK_>::ELEMS];
    // ragged batches (m_dev): the tile list is the EFFECTIVE one -- the XCD remap deals contiguous chunks of it to the
    // XCDs, so remapping the upper-bound list would leave the XCDs that own the empty tail idle
    const GemmDev q = effective<AK>(p);
    const unsigned pfv = prefetch_lines(p);
    const int gx = gridDim.x;
    const int gy = AK ? (int)gridDim.y : min((int)gridDim.y, (q.M + BM - 1) / BM);
    const int orig = blockIdx.y * gx + blockIdx.x;
    if (orig < gx * gy) {
        const int t = xcd_remap(orig, gx * gy);
        int by, bx;
        tile_coords(t, gx, gy, q.xcs, by, bx);
        gemm_body<T, BM, BN, AK, BK_, PF2, 0, WIDE>(q, bx, by, blockIdx.z, sA, sB);
    }
    asm volatile("" :: "v"(pfv));
}

// Skinny products (M <= 64: the 2-token decode step, poolers, classifier heads): the weight matrix is read once
// and nothing is reused inside a workgroup, so there is no LDS staging -- every wave loads its MFMA fragments
// straight from global memory (16 B per lane, k-contiguous rows).  Workgroup = 16 output columns x all rows;
// its 8 waves split K (3 k-blocks in flight per wave: the loop is latency bound, so memory-level parallelism is
// what matters), partial accumulators meet in LDS, wave i < 4 finishes row tile i.
constexpr int SKINNY_WAVES = 8, SKINNY_UNROLL = 3;
// ARGMAX (greedy decoding, model.py:896-900): instead of storing the logits, every workgroup reduces its 16
// columns to (max, first index of the max) per row -> part_val/part_idx [M][gridDim.x]; argmax_parts_kernel
// finishes the rows.  The maximum is taken over the f32 accumulators (+ bias).
struct ArgmaxOut { float* part_val; int* part_idx; };
// Weight fragments of the skinny products: every workgroup reads ITS 16 (or 128) weight rows exactly once, and in decoding the
// 217 MB of weights + ~150 MB of K / V rows per step cycle through a 256 MB Infinity Cache -- non-temporal loads
// (MI355X_MICROARCH.md "nt-weights": once-read streamed weights, 5-10 % per decode layer).  -DMVLT_SKINNY_NT=0 restores the
// default cache policy (A/B builds).
#ifndef MVLT_SKINNY_NT
#define MVLT_SKINNY_NT 1
#endif
template <typename F> MVLT_DEV F skinny_wload(const F* q) {
#if MVLT_SKINNY_NT
    return __builtin_nontemporal_load(q);
#else
    return *q;
#endif
}
template <typename T, bool ARGMAX = false>
__global__ __launch_bounds__(64 * SKINNY_WAVES) void gemm_skinny_kernel(const GemmDev p_in, const ArgmaxOut am) {
    const GemmDev p = effective<false>(p_in);          // ragged row counts (m_dev): rows beyond it are neither read nor written
    if (p.M <= 0) return;
    using M_ = Mma<T>;
    using Frag = typename M_::Frag;
    constexpr int KB = M_::KB, E = TypeInfo<T>::E;
    __shared__ f32x4 red[SKINNY_WAVES][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r15 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const T* A = reinterpret_cast<const T*>(p.A);
    const T* B = reinterpret_cast<const T*>(p.B);
    const T* brow = B + (long)min(n0 + r15, p.N - 1) * p.ldb + g * E;
    const T* arow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) arow[i] = A + (long)min(16 * i + r15, p.M - 1) * p.lda + g * E;
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the finishing waves fetch their bias / residual values now, so the epilogue does not start with a
    // dependent memory round trip (the decode step is a chain of ~100 such kernels)
    constexpr int FASTMASK = MVLT_EPI_BIAS | MVLT_EPI_GELU | MVLT_EPI_RESIDUAL;
    const bool fast_epi = !ARGMAX && (p.epi & ~FASTMASK) == 0 && p.epi_vec && n0 + 16 <= p.N && wave < 4 && 16 * wave + r15 < p.M;
    f32x4 pre_bias{0.f, 0.f, 0.f, 0.f}, pre_res{0.f, 0.f, 0.f, 0.f};
    if (fast_epi) {
        if (p.epi & MVLT_EPI_BIAS) pre_bias = *reinterpret_cast<const f32x4*>(p.bias + n0 + 4 * g);
        if (p.epi & MVLT_EPI_RESIDUAL)
            pre_res = load4f(reinterpret_cast<const T*>(p.residual) + (long)(16 * wave + r15) * p.ldr + n0 + 4 * g);
    }
    const int nkb = p.K / KB;
    for (int kb0 = wave; kb0 < nkb; kb0 += SKINNY_WAVES * SKINNY_UNROLL) {
        Frag fb[SKINNY_UNROLL], fa[SKINNY_UNROLL][4];
#pragma unroll
        for (int u = 0; u < SKINNY_UNROLL; ++u) {
            const int kb = kb0 + u * SKINNY_WAVES;
            const int k = (kb < nkb ? kb : kb0) * KB;          // past the end: reload a valid block, never multiplied
            fb[u] = skinny_wload(reinterpret_cast<const Frag*>(brow + k));
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[u][i] = *reinterpret_cast<const Frag*>(arow[i] + k);
        }
#pragma unroll
        for (int u = 0; u < SKINNY_UNROLL; ++u) {
            if (kb0 + u * SKINNY_WAVES < nkb) {
#pragma unroll
                for (int i = 0; i < 4; ++i) M_::mma(acc[i], fb[u], fa[u][i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave][i][lane] = acc[i];
    __syncthreads();
    const int i = wave;                                  // row tile finished by this wave
    if (i < 4 && 16 * i < p.M) {
        f32x4 v = red[0][i][lane];
#pragma unroll
        for (int w = 1; w < SKINNY_WAVES; ++w) v += red[w][i][lane];
        // acc[r] <-> n = n0 + 4*g + r, m = 16*i + (lane & 15)   (same orientation as gemm_body)
        if (!ARGMAX) {
            if (fast_epi) {
                f32x4 o = v + pre_bias;
                if (p.epi & MVLT_EPI_GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = gelu_f(o[r]);
                }
                o += pre_res;
                store4f(reinterpret_cast<T*>(p.C) + (long)(16 * i + r15) * p.ldc + n0 + 4 * g, o);
            } else {
                epilogue4<T>(p, 16 * i + r15, n0 + 4 * g, v);
            }
        } else {
            float best = -3.0e38f; int bi = 0x7fffffff;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + 4 * g + r;
                if (n < p.N) {
                    const float x = v[r] + ((p.epi & MVLT_EPI_BIAS) ? p.bias[n] : 0.f);
                    if (x > best) { best = x; bi = n; }          // ascending n: ties keep the first index
                }
            }
#pragma unroll
            for (int o = 16; o < 64; o <<= 1) {
                const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
                if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            const int m = 16 * i + r15;
            if (g == 0 && m < p.M) { am.part_val[(long)m * gridDim.x + blockIdx.x] = best; am.part_idx[(long)m * gridDim.x + blockIdx.x] = bi; }
        }
    }
}

// Skinny product with K split over workgroups (decode: M = 2B rows, N = 768, K = 768 / 3072).  The plain skinny kernel
// gives every 16-column workgroup the WHOLE activation matrix to read (N/16 x M x K bytes through L2: 48 x 393 KB for
// the FFN-out product, 10 us per workgroup at the ~50 GB/s a CU takes in); here workgroup (j, s) reads only k-slice s
// of it and writes its partial 64x16 tile into the slab of its slice.  No epilogue: the consumer
// (mvlt_layernorm_acc_fwd: sum of the slices + bias + residual, LayerNorm) is the launch that follows anyway.
template <typename T>
__global__ __launch_bounds__(64 * SKINNY_WAVES) void gemm_skinny_accum_kernel(const GemmDev p, float* accout) {
    using M_ = Mma<T>;
    using Frag = typename M_::Frag;
    constexpr int KB = M_::KB, E = TypeInfo<T>::E;
    __shared__ f32x4 red[SKINNY_WAVES][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r15 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const T* A = reinterpret_cast<const T*>(p.A);
    const T* B = reinterpret_cast<const T*>(p.B);
    const T* brow = B + (long)min(n0 + r15, p.N - 1) * p.ldb + g * E;
    const T* arow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) arow[i] = A + (long)min(16 * i + r15, p.M - 1) * p.lda + g * E;
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nkb = p.K / KB;
    const int per = (nkb + (int)gridDim.y - 1) / (int)gridDim.y;
    const int kb_lo = blockIdx.y * per, kb_hi = min(nkb, kb_lo + per);
    for (int kb0 = kb_lo + wave; kb0 < kb_hi; kb0 += SKINNY_WAVES * SKINNY_UNROLL) {
        Frag fb[SKINNY_UNROLL], fa[SKINNY_UNROLL][4];
#pragma unroll
        for (int u = 0; u < SKINNY_UNROLL; ++u) {
            const int kb = kb0 + u * SKINNY_WAVES;
            const int k = (kb < kb_hi ? kb : kb0) * KB;
            fb[u] = skinny_wload(reinterpret_cast<const Frag*>(brow + k));
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[u][i] = *reinterpret_cast<const Frag*>(arow[i] + k);
        }
#pragma unroll
        for (int u = 0; u < SKINNY_UNROLL; ++u) {
            if (kb0 + u * SKINNY_WAVES < kb_hi) {
#pragma unroll
                for (int i = 0; i < 4; ++i) M_::mma(acc[i], fb[u], fa[u][i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave][i][lane] = acc[i];
    __syncthreads();
    const int i = wave;
    if (i < 4 && 16 * i + r15 < p.M) {
        f32x4 v = red[0][i][lane];
#pragma unroll
        for (int w = 1; w < SKINNY_WAVES; ++w) v += red[w][i][lane];
        // slab of this k-slice: [gridDim.y][M][N] f32, written once with plain stores; the consumer (ln_acc_fwd_kernel) adds the
        // slices in slice order -- no float atomics, no zeroing pass, bit-reproducible (round 5; the atomic form cost ~1 us more)
        float* c = accout + ((long)blockIdx.y * p.M + 16 * i + r15) * p.N + n0 + 4 * g;
        if (n0 + 4 * g + 4 <= p.N) store4f(c, v);
        else
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n0 + 4 * g + r < p.N) c[r] = v[r];
    }
}

// one wave per row: (max, first index) over the workgroup partials
__global__ __launch_bounds__(64) void argmax_parts_kernel(const float* part_val, const int* part_idx, int nparts,
                                                          int64_t* out_idx, float* out_val) {
    const long base = (long)blockIdx.x * nparts;
    float best = -3.0e38f; int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < nparts; i += 64) {
        const float v = part_val[base + i]; const int idx = part_idx[base + i];
        if (v > best || (v == best && idx < bi)) { best = v; bi = idx; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (threadIdx.x == 0) { out_idx[blockIdx.x] = bi; if (out_val) out_val[blockIdx.x] = best; }
}

// Decoder GEMM of the last-row MLM head fused with the greedy pick, wide form (round 5): a workgroup = 128 vocabulary columns,
// wave w = columns 16 w .. +16 over the WHOLE reduction -- no k-split across waves, so no LDS reduction and no barrier; the
// 47 MB weight matrix is streamed once by 239 workgroups (one per CU) with UNR k-blocks in flight per wave, the 32-64
// activation rows come from L1 / L2.  The 16-column form above launches 1,908 workgroups that each re-read the whole
// activation matrix and meet in LDS: 29 us for a product whose bytes take 9 us.  Same partial layout (one (max, index) per row
// and 16 columns), so the finishing kernel is shared.
template <typename T, int NRT>
__global__ __launch_bounds__(64 * SKINNY_WAVES) void gemm_argmax128_kernel(const GemmDev p, const ArgmaxOut am, int nparts) {
    using M_ = Mma<T>;
    using Frag = typename M_::Frag;
    constexpr int KB = M_::KB, E = TypeInfo<T>::E, UNR = NRT <= 2 ? 12 : 8;          // k-blocks in flight per wave (12 KB of weights; K = 768 in two rounds)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), r15 = lane & 15, g = lane >> 4;
    const int n0 = (blockIdx.x * SKINNY_WAVES + wave) * 16;
    if (n0 >= p.N) return;
    const T* A = reinterpret_cast<const T*>(p.A);
    const T* brow = reinterpret_cast<const T*>(p.B) + (long)min(n0 + r15, p.N - 1) * p.ldb + g * E;
    const T* arow[NRT];
#pragma unroll
    for (int i = 0; i < NRT; ++i) arow[i] = A + (long)min(16 * i + r15, p.M - 1) * p.lda + g * E;
    f32x4 acc[NRT];
#pragma unroll
    for (int i = 0; i < NRT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nkb = p.K / KB;
    for (int kb0 = 0; kb0 < nkb; kb0 += UNR) {
        Frag fb[UNR], fa[UNR][NRT];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int k = min(kb0 + u, nkb - 1) * KB;          // past the end: reload the last block, never multiplied
            fb[u] = skinny_wload(reinterpret_cast<const Frag*>(brow + k));
#pragma unroll
            for (int i = 0; i < NRT; ++i) fa[u][i] = *reinterpret_cast<const Frag*>(arow[i] + k);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (kb0 + u < nkb) {
#pragma unroll
                for (int i = 0; i < NRT; ++i) M_::mma(acc[i], fb[u], fa[u][i]);
            }
        }
    }
    // acc[i][r] <-> n = n0 + 4 g + r, m = 16 i + r15
    f32x4 bias4{0.f, 0.f, 0.f, 0.f};
    if (p.epi & MVLT_EPI_BIAS) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bias4[r] = p.bias[min(n0 + 4 * g + r, p.N - 1)];
    }
    const int part = blockIdx.x * SKINNY_WAVES + wave;
#pragma unroll
    for (int i = 0; i < NRT; ++i) {
        float best = -3.0e38f; int bi = 0x7fffffff;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + 4 * g + r;
            const float x = acc[i][r] + bias4[r];
            if (n < p.N && x > best) { best = x; bi = n; }          // ascending n: ties keep the first index
        }
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {
            const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        const int m = 16 * i + r15;
        if (g == 0 && m < p.M) { am.part_val[(long)m * nparts + part] = best; am.part_idx[(long)m * nparts + part] = bi; }
    }
}

// Finish of the greedy pick for ALL rows in one workgroup, with the per-token bookkeeping of greedy_search (model.py:896-913)
// folded in: next = argmax; finished samples emit PAD; unfinished &= (next != EOS); ids[:, col] = next; scores[:, col] = max
// logit; new_ids[:, 0] = next (the first of the two tokens the next cached step feeds); alive[col] = any sample unfinished;
// past += 1 (the cache position the NEXT forward reads: the previous step's [MASK] slot is overwritten); col += 1.
// Replaces argmax_parts_kernel + ten one-line torch kernels per replayed decode step.  M <= 64; 16 waves, wave w = rows w, w + 16, ...
struct GreedyState {
    int64_t* unfinished; int64_t eos, pad; int has_eos;
    int64_t* col; int32_t* past;
    int64_t* ids; long ld_ids; float* scores; long ld_scores; int64_t* alive; int64_t* new_ids; long ld_new;
    int32_t* ticket;
};
// One workgroup per row (256 threads: ~8 partials per thread, one memory round trip; a single workgroup walking all rows took
// 26 us).  alive[col] is raised with an atomic max by the rows that are still unfinished (the caller zeroes `alive` when a
// decode starts); the LAST workgroup to arrive (ticket) advances col and past and re-arms the ticket.  Every workgroup reads
// col before it draws its ticket, and col is written only after all tickets are drawn.
__global__ __launch_bounds__(256) void greedy_pick_kernel(const float* part_val, const int* part_idx, int nparts, int M, const GreedyState st) {
    __shared__ float s_val[4];
    __shared__ int s_idx[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = blockIdx.x;
    const long col = st.col[0];
    const long base = (long)m * nparts;
    float best = -3.0e38f; int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < nparts; i += 256) {
        const float v = part_val[base + i]; const int idx = part_idx[base + i];
        if (v > best || (v == best && idx < bi)) { best = v; bi = idx; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) { s_val[wave] = best; s_idx[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float ov = s_val[w]; const int oi = s_idx[w];
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        int64_t nxt = bi;
        if (st.has_eos) {
            const int64_t unf = st.unfinished[m];
            nxt = nxt * unf + st.pad * (1 - unf);
            const int64_t unf2 = unf * (nxt != st.eos ? 1 : 0);
            st.unfinished[m] = unf2;
            if (unf2) atomicMax(reinterpret_cast<unsigned long long*>(st.alive + col), 1ULL);
        }
        st.ids[(long)m * st.ld_ids + col] = nxt;
        st.scores[(long)m * st.ld_scores + col] = best;
        st.new_ids[(long)m * st.ld_new] = nxt;
        __threadfence();
        if (atomicAdd(st.ticket, 1) == M - 1) {
            *st.ticket = 0;
            st.col[0] = col + 1;
            if (st.past) st.past[0] += 1;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight-gradient tile with BOTH operands k-major through LDS-DMA (round 6): dW[m, n] = sum_k dY[k, m] X[k, n].
// The register-staged form (gemm_body, DEEP = 2) keeps two register sets of the next tiles beside 64 accumulators: 243 VGPRs
// at 128 x 128, i.e. two waves per SIMD and NOTHING else resident on a CU that holds two such workgroups
// (profiles/r6_ln_bwd.md).  Here the tiles go global -> LDS by LDS-DMA in the k-major image the transposing fragment reads
// expect ([64 k][R] with the 32-byte units XOR-swizzled by kswz<R>(k), the swizzle applied to the per-lane SOURCE address
// exactly as gemm_glds_kernel<.., BKM> does for its B tile): no staging registers, no ds_write pass.
//   * K tail / ragged reductions (m_dev): k-rows at or beyond the reduction length are fetched from a 256-byte page of
//     zeros instead of the operand (a per-lane pointer select per request), for BOTH operands (0 x NaN of an unwritten
//     tail row would poison the sum);
//   * bias gradient db[m] = sum_k dY[k, m]: on the matrix pipe, dY^T . 1 (one extra MFMA per A fragment and k-block in the
//     wn = 0 waves of the first column tile of every tile row; accumulator column 0 holds the sum) -- the staging registers
//     the register form sums from do not exist here;
//   * two LDS stages, one barrier per k-tile, counted by hand (the DMA is inline asm: gemm_dev.h).
__device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];          // zero-initialised device memory

template <int BM, int BN, int NST>
MVLT_DEV void wgrad_glds_tile(const GemmDev& p, const int bx, const int by, bf16_t* smem) {
    using T = bf16_t;
    constexpr int BKE = 64, FM = BM / 32, FN = BN / 32;
    static_assert((BM == 64 || BM == 128) && (BN == 64 || BN == 128), "k-major LDS-DMA tiles: 64 or 128 wide");
    const int m0 = by * BM, n0 = bx * BN;
    const int ke = p.K;
    const int nkt = (ke + BKE - 1) / BKE;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wm = wave >> 1, wn = wave & 1;
    const T* A = reinterpret_cast<const T*>(p.A);
    const T* B = reinterpret_cast<const T*>(p.B);
    constexpr int STAGE = (BM + BN) * BKE;            // elements per LDS stage: A image [64][BM], then B image [64][BN]
    const T* srcA[BM / 32];
    const T* srcB[BN / 32];
    int krowA[BM / 32], krowB[BN / 32];
#pragma unroll
    for (int j = 0; j < BM / 32; ++j) {
        constexpr int KPI = 512 / BM, XM = BM / 8 - 1;
        const int krow = (wave * (BM / 32) + j) * KPI + lane / (BM / 8), x = lane & XM;
        const int ch = (((x >> 1) ^ kswz<BM>(krow)) << 1) | (x & 1);
        krowA[j] = krow;
        srcA[j] = A + (long)krow * p.lda + min(m0 + ch * 8, max(p.M - 8, 0));
    }
#pragma unroll
    for (int j = 0; j < BN / 32; ++j) {
        constexpr int KPI = 512 / BN, XM = BN / 8 - 1;
        const int krow = (wave * (BN / 32) + j) * KPI + lane / (BN / 8), x = lane & XM;
        const int ch = (((x >> 1) ^ kswz<BN>(krow)) << 1) | (x & 1);
        krowB[j] = krow;
        srcB[j] = B + (long)krow * p.ldb + min(n0 + ch * 8, max(p.N - 8, 0));
    }
    const T* zero = reinterpret_cast<const T*>(g_zero_page) + (lane & 15) * 8;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) T*)smem;
    auto fill = [&](int stage, int kt) {
        const unsigned sa = lds0 + (unsigned)(stage * STAGE) * 2u, sb = sa + (unsigned)(BM * BKE) * 2u;
        const int k0 = kt * BKE;
#pragma unroll
        for (int j = 0; j < BM / 32; ++j)
            glds16_asm(k0 + krowA[j] < ke ? srcA[j] + (long)k0 * p.lda : zero, sa + (wave * (BM / 32) + j) * 1024);
#pragma unroll
        for (int j = 0; j < BN / 32; ++j)
            glds16_asm(k0 + krowB[j] < ke ? srcB[j] + (long)k0 * p.ldb : zero, sb + (wave * (BN / 32) + j) * 1024);
    };
    f32x4 acc[FM][FN], cacc[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        cacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const bool do_colsum = p.a_colsum != nullptr && bx == 0 && wn == 0;          // (wave-uniform)
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;
    static_assert(NST >= 2 && NST <= 4, "two to four LDS stages (four measured slower than three: not instantiated)");
    constexpr int PER = BM / 32 + BN / 32;                     // LDS-DMA requests per wave and k-tile
#pragma unroll
    for (int st = 0; st < NST - 1; ++st) if (st < nkt) fill(st, st);
    for (int kt = 0; kt < nkt; ++kt) {
        // this wave's share of tile kt has landed (NST stages: the requests of the next NST - 2 tiles may still be in flight -- counted)
        const int ahead = min(NST - 2, nkt - 1 - kt);
        if (NST >= 4 && ahead >= 2) wait_vmcnt<(NST >= 4 ? 2 : 0) * PER>();
        else if (NST >= 3 && ahead == 1) wait_vmcnt<(NST >= 3 ? 1 : 0) * PER>();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                       // everybody's has; everybody is done reading tile kt - 1
        if (kt + NST - 1 < nkt) fill((kt + NST - 1) % NST, kt + NST - 1);
        const T* a = smem + (kt % NST) * STAGE;
        const T* b = a + BM * BKE;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            Mma<T>::Frag fa[FM], fb[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) fa[i] = tile_frag<T, BM, true>(a, wm * (BM / 2) + i * 16, kb);
#pragma unroll
            for (int j = 0; j < FN; ++j) fb[j] = tile_frag<T, BN, true>(b, wn * (BN / 2) + j * 16, kb);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) Mma<T>::mma(acc[i][j], fb[j], fa[i]);
            if (do_colsum) {
#pragma unroll
                for (int i = 0; i < FM; ++i) Mma<T>::mma(cacc[i], ones, fa[i]);
            }
        }
    }
    if (do_colsum && lane < 16) {
        // cacc[i][r] = sum_k A[k, m] for m = m0 + wm BM/2 + 16 i + lane (every r: the first operand's rows are all ones)
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = m0 + wm * (BM / 2) + i * 16 + lane;
            if (m < p.M) p.a_colsum[m] = (p.epi & MVLT_EPI_ACCUM) ? p.a_colsum[m] + cacc[i][0] : cacc[i][0];
        }
    }
    tile_epilogue<T, FM, FN>(p, m0 + wm * (BM / 2), n0 + wn * (BN / 2), acc);
    __syncthreads();                                           // (persistent callers refill stage 0)
}

// Several independent products in one launch (the weight gradients of one layer): the tile lists of the
// items are concatenated, a workgroup finds its item by a scan of the (<= 8) prefix counts.
constexpr int GROUP_MAX = 8;
struct GemmGroupDev { int n; int split; int start[GROUP_MAX + 1]; GemmDev g[GROUP_MAX]; };   // start[]: in (tile, k-slice) units

template <typename T, int BM, int BN, bool AK, bool BK_, int DEEP>
__global__ __launch_bounds__(256, 2) void gemm_group_kernel(const GemmGroupDev gp) {
    __shared__ __attribute__((aligned(16))) T sA[(DEEP == 2 ? 2 : 1) * TileGeom<T, BM, AK>::ELEMS];
    __shared__ __attribute__((aligned(16))) T sB[(DEEP == 2 ? 2 : 1) * TileGeom<T, BN, BK_>::ELEMS];
    // persistent: the grid may be smaller than the tile list (the launcher caps the workgroups per CU so the
    // dgrad chain on the main stream keeps most of every CU); gemm_body ends on a barrier, so LDS is reusable
    const int total = gp.start[gp.n];
    for (int t0 = blockIdx.x; t0 < total; t0 += gridDim.x) {
        const int t = xcd_remap(t0, total);
        int i = 0;
        while (i + 1 < gp.n && t >= gp.start[i + 1]) ++i;
        const GemmDev p = effective<AK>(gp.g[i]);
        const int gx = (p.N + BN - 1) / BN, gy = (p.M + BM - 1) / BM;
        const int local = t - gp.start[i];
        const int bz = local / (gx * gy), tile = local - bz * gx * gy;          // k-slice bz of output tile `tile`
        // the list runs along the SHORTER side of the tile grid first: the contiguous chunk of it that lands on one XCD (xcd_remap)
        // is then a block, not a sliver of a few rows x all columns -- a [768, 3072] gradient: 6 + 9 operand bands per L2 instead of 3 + 24
        int by, bx;
        if (gx > gy) { bx = tile / gy; by = tile - bx * gy; }
        else { by = tile / gx; bx = tile - by * gx; }
        gemm_body<T, BM, BN, AK, BK_, false, DEEP>(p, bx, by, bz, sA, sB);
    }
}

// the same list walked by the LDS-DMA tile (bf16, both operands k-major, no k-slices): ~150 instead of 243 VGPRs at 128 x 128
template <int BM, int BN, int NST>
__global__ __launch_bounds__(256, 2) void gemm_group_glds_kernel(const GemmGroupDev gp) {
    constexpr bool DYN = NST * (BM + BN) * 64 * 2 > 65536;
    extern __shared__ __attribute__((aligned(16))) char wg_dyn[];
    __shared__ __attribute__((aligned(16))) bf16_t wg_static[DYN ? 8 : NST * (BM + BN) * 64];
    bf16_t* const smem = DYN ? reinterpret_cast<bf16_t*>(wg_dyn) : wg_static;
    const int total = gp.start[gp.n];
    for (int t0 = blockIdx.x; t0 < total; t0 += gridDim.x) {
        const int t = xcd_remap(t0, total);
        int i = 0;
        while (i + 1 < gp.n && t >= gp.start[i + 1]) ++i;
        const GemmDev p = effective<true>(gp.g[i]);
        const int gx = (p.N + BN - 1) / BN, gy = (p.M + BM - 1) / BM;
        const int tile = t - gp.start[i];
        int by, bx;
        if (gx > gy) { bx = tile / gy; by = tile - bx * gy; }
        else { by = tile / gx; bx = tile - by * gx; }
        wgrad_glds_tile<BM, BN, NST>(p, bx, by, smem);
    }
}

// slabs -> output: 64 output quads per block, the slabs are shared out over 4 waves and combined in LDS
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmDev p_in) {
    __shared__ f32x4 red[4][64];
    GemmDev p = p_in;
    // m_dev counts the rows of a k-contiguous A (= rows of the output): slab rows beyond it were never written.  For a k-major
    // A (weight gradients) it is the reduction length and the output keeps all M rows.
    if (p.m_dev && !p.a_kmajor) p.M = min(p.M, max(*p.m_dev, 0));
    const int nq = (p.N + 3) / 4;
    const long total = (long)p.M * nq;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (long base = (long)blockIdx.x * 64; base < total; base += (long)gridDim.x * 64) {
        const long idx = base + lane;
        const bool ok = idx < total;
        const int m = ok ? (int)(idx / nq) : 0, n = ok ? (int)(idx % nq) * 4 : 0;
        f32x4 v{0.f, 0.f, 0.f, 0.f};
        if (ok) {
            if ((p.N & 3) == 0) {
                // eight slabs per round trip (a plain loop waits out one load latency per slab: 30 us for the 96 slices of
                // the patch-embedding weight gradient, at the very end of the backward pass), summed in slice order
                const long sstride = (long)p.M * p.N;
                const float* ws0 = p.ws + (long)m * p.N + n;
                for (int s = w; s < p.split_k; s += 32) {
                    f32x4 t[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int sj = s + 4 * j;
                        t[j] = sj < p.split_k ? load4f(ws0 + sj * sstride) : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) v += t[j];
                }
            } else {
                for (int s = w; s < p.split_k; s += 4) {
                    const float* ws = p.ws + ((long)s * p.M + m) * p.N + n;
                    for (int r = 0; r < 4; ++r) if (n + r < p.N) v[r] += ws[r];
                }
            }
        }
        red[w][lane] = v;
        __syncthreads();
        if (w == 0 && ok) epilogue4<T>(p, m, n, red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
        __syncthreads();
    }
    if (p.a_colsum) {
        for (int m = blockIdx.x * 256 + threadIdx.x; m < p.M; m += gridDim.x * 256) {
            float s0 = 0.f;
            for (int z = 0; z < p.split_k; z += 8) {
                float t[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = z + j < p.split_k ? p.ws_colsum[(long)(z + j) * p.M + m] : 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) s0 += t[j];
            }
            p.a_colsum[m] = (p.epi & MVLT_EPI_ACCUM) ? p.a_colsum[m] + s0 : s0;
        }
    }
}

template <typename T, int BM, int BN, bool PF2>
int launch_layout2(const GemmDev& d, bool ak, bool bk, dim3 grid, hipStream_t s) {
    if constexpr (sizeof(T) == 2) {          // bf16 rows out, 16 bytes per lane (GemmDev.wide): forward / dgrad products only
        if (d.wide && !ak) {
            if (bk) hipLaunchKernelGGL((gemm_kernel<T, BM, BN, false, true, PF2, true>), grid, dim3(256), 0, s, d);
            else hipLaunchKernelGGL((gemm_kernel<T, BM, BN, false, false, PF2, true>), grid, dim3(256), 0, s, d);
            return 0;
        }
    }
    if (!ak && !bk) hipLaunchKernelGGL((gemm_kernel<T, BM, BN, false, false, PF2>), grid, dim3(256), 0, s, d);
    else if (!ak && bk) hipLaunchKernelGGL((gemm_kernel<T, BM, BN, false, true, PF2>), grid, dim3(256), 0, s, d);
    else if (ak && bk) hipLaunchKernelGGL((gemm_kernel<T, BM, BN, true, true, PF2>), grid, dim3(256), 0, s, d);
    else hipLaunchKernelGGL((gemm_kernel<T, BM, BN, true, false, PF2>), grid, dim3(256), 0, s, d);
    return 0;
}
template <typename T, int BM, int BN>
int launch_layout(const GemmDev& d, bool ak, bool bk, dim3 grid, hipStream_t s) {
    // deeper prefetch only for the small tiles (cheap in registers) and long reductions
    const int bke = 128 / (int)sizeof(T);
    const bool pf2 = BM == 64 && BN == 64 && d.k_per_split >= 12 * bke;
    if (pf2) return launch_layout2<T, BM, BN, (BM == 64 && BN == 64)>(d, ak, bk, grid, s);
    return launch_layout2<T, BM, BN, false>(d, ak, bk, grid, s);
}

struct Plan { int bm, bn, split; };

// 160 x 128 tiles exist only as the LDS-DMA kernel with the 16-byte epilogue (gemm_glds_kernel<160, 128, *, true>): every
// condition of that path is checked here, at PLAN time (the grid is sized for the planned tile)
bool tile160_ok(const MvltGemm* p) {
    const int epi = p->epilogue;
    if (p->dtype != MVLT_BF16 || p->a_kmajor || p->split_k > 1 || p->a_colsum) return false;
    if (p->K % 64 || p->N % 8 || p->lda % 8 || p->ldb % 8 || p->ldc % 8) return false;
    if (!aligned16(p->A) || !aligned16(p->B) || !aligned16(p->C)) return false;
    if (epi & (MVLT_EPI_OUT_F32 | MVLT_EPI_ACCUM)) return false;
    if ((epi & MVLT_EPI_RESIDUAL) && (p->ldr % 8 || !aligned16(p->residual))) return false;
    if ((epi & MVLT_EPI_SAVE_PRE) && !aligned16(p->pre)) return false;
    if ((epi & MVLT_EPI_MUL_GELU_GRAD) && !aligned16(p->aux)) return false;
    if ((epi & MVLT_EPI_BIAS) && !aligned16(p->bias)) return false;
    return true;
}

Plan choose_plan(const MvltGemm* p) {
    // tile: BN = 128 when N is a multiple of 128 (or large), else 96 (all model
    // widths are multiples of 96), 64 for tiny N.  BM = 128 unless that leaves
    // the 256 CUs under-filled.  split-K only when requested (>1) or auto (0).
    Plan pl;
    pl.bn = (p->N % 128 == 0) ? 128 : (p->N % 96 == 0 || p->N > 512) ? 96 : (p->N <= 64 ? 64 : (p->N <= 96 ? 96 : 128));
    if (p->N > 512 && p->N % 128 != 0 && p->N % 96 != 0) pl.bn = 128;
    long tiles128 = (long)ceil_div(p->M, 128) * ceil_div(p->N, pl.bn);
    pl.bm = (tiles128 >= 384 || p->M > 64 * 1024) ? 128 : 64;
    if (pl.bn == 64) pl.bm = 64;
    // narrow outputs (N = C of a Swin stage / 768): 64x128 tiles leave most of the 1024 resident slots
    // empty; 64x64 tiles quadruple the workgroup count (measured 10-25 % faster on those shapes)
    if (pl.bm == 64 && pl.bn == 128 && p->N % 64 == 0 &&
        (long)ceil_div(p->M, 64) * ceil_div(p->N, 128) < 512 && !(p->a_kmajor && p->b_kmajor)) pl.bn = 64;
    // wide outputs with a short reduction (BertLayer qkv / FFN-in forward, FFN-out dgrad: N >= 2304, K = 768): 64 x 128 tiles
    // put four to five workgroups on a CU instead of three and hide more of the operand latency: 27.6 vs 29.4 us stand-alone
    // on 3090 x 3072 x 768, +0.35 % pairs/s in the step (5 interleaved runs each)
    if (pl.bm == 128 && pl.bn == 128 && p->N >= 2304 && p->K <= 1024 && !(p->a_kmajor && p->b_kmajor)) pl.bm = 64;
    // forward products (x W^T) of the stage-2 / BertLayer size with a short reduction: 64 x 64 tiles are as fast as the wide ones
    // stand-alone (6272 x 1536 x 384: 19.8 vs 20.3 us, 3090 x 2304 x 768: 22.9 vs 23.1) and finish more evenly inside the forward
    // pass, a serial chain of ~250 launches where every tail is exposed: +0.55 % pairs/s (6 interleaved runs each)
    if (!p->a_kmajor && !p->b_kmajor && p->N % 64 == 0 && p->M <= 8192 && p->K <= 768 && p->N <= 2304) { pl.bm = 64; pl.bn = 64; }
    // ONE round of 160 x 128 tiles (two workgroups per CU: 512 slots) where that fills 2/3 .. all of the chip: 71 FLOP per byte of
    // L2 -> LDS traffic against 32-43 for the 64-row tiles (stand-alone 3150 x 3072 x 768 + GELU 27.1 vs 29.9 us, 3150 x 2304 x 768
    // 21.5 vs 23.7, 6272 x 1536 x 384 22.1 vs 24.5, 4192 x 2304 x 768 22.0 vs 28.3; scripts/tile160_sweep.sh)
    // Forward products only: for the dgrads of the same shapes the 72-KB workgroups share a CU badly with the weight-gradient stream
    // (step 11.66 -> 11.90 ms with them, 11.71 -> 11.64 ms with the forward products alone; profiles/r5_tile160_ab.txt).
    static const bool t160 = [] { const char* e = getenv("MVLT_TILE160"); return !e || atoi(e) != 0; }();
    if (t160 && p->N % 128 == 0 && p->N >= 1536 && p->K <= 1024 && !p->b_kmajor && tile160_ok(p)) {
        const long t = (long)ceil_div(p->M, 160) * (p->N / 128);
        // ragged batches (m_dev): M is the dense upper bound; the tiles beyond the packed row count leave at once, so the EFFECTIVE
        // count is ~3/4 of t (BertLayer FFN-in: 648 launched, ~480 with work).  Round 6: accepted up to 680 launched tiles -- at
        // worst (every caption at full length) that is a second, quarter-full round: 2 x 13.5 us against 33.6 us on 64 x 128 tiles.
        static const bool ragged160 = [] { const char* e = getenv("MVLT_TILE160_RAGGED"); return !e || atoi(e) != 0; }();
        const long tmax = (p->m_dev && ragged160) ? 680 : 512;
        if (t > 340 && t <= tmax) { pl.bm = 160; pl.bn = 128; }
    }
    if (const char* ov = getenv("MVLT_TILE")) {          // experiments: MVLT_TILE=bm,bn
        int a = 0, b = 0;
        if (sscanf(ov, "%d,%d", &a, &b) == 2 && (a == 128 || a == 64) && (b == 128 || b == 96 || b == 64) &&
            !(a == 128 && b == 64)) { pl.bm = a; pl.bn = b; }
        if (a == 160 && b == 128 && tile160_ok(p)) { pl.bm = 160; pl.bn = 128; }
    }
    long tiles = (long)ceil_div(p->M, pl.bm) * ceil_div(p->N, pl.bn);
    int split = p->split_k;
    if (split == 0) {
        split = 1;
        const int bke = (p->dtype == MVLT_BF16) ? 64 : 32;
        const int nkt = ceil_div(p->K, bke);
        if (tiles < 200 && nkt >= 16 && pl.bm != 160) {          // small outputs with a long reduction only (wgrads)
            split = (int)((768 + tiles - 1) / tiles);
            if (split > nkt / 8) split = nkt / 8;
            if (split > 96) split = 96;
            if (split < 1) split = 1;
        }
    }
    pl.split = split;
    return pl;
}

}  // namespace

// M <= 64, both operands k-contiguous, whole k-blocks, no split requested -> gemm_skinny_kernel
template <typename T>
static bool is_skinny(const MvltGemm* p) {
    return p->M <= 64 && !p->a_kmajor && !p->b_kmajor && p->K % Mma<T>::KB == 0 && p->split_k <= 1 && !p->a_colsum;
}

static bool is_skinny_any(const MvltGemm* p) {
    return p->dtype == MVLT_BF16 ? is_skinny<bf16_t>(p) : (p->dtype == MVLT_F32 && is_skinny<float>(p));
}

extern "C" size_t mvlt_gemm_workspace_bytes(const MvltGemm* p) {
    if (!p) return 0;
    if (is_skinny_any(p)) return 0;
    Plan pl = choose_plan(p);
    return pl.split > 1 ? (size_t)pl.split * p->M * ((size_t)p->N + 1) * sizeof(float) : 0;
}

extern "C" int mvlt_gemm_plan(const MvltGemm* p, int* bm, int* bn, int* split_k) {
    if (!p || !bm || !bn || !split_k) return MVLT_ERR_ARG;
    if (is_skinny_any(p)) {      // (alignment permitting)
        *bm = 64; *bn = 16; *split_k = 1;
        return MVLT_OK;
    }
    Plan pl = choose_plan(p);
    *bm = pl.bm; *bn = pl.bn; *split_k = pl.split;
    return MVLT_OK;
}

// MvltGemm -> kernel argument block for a given plan
template <typename T>
static int fill_dev(const MvltGemm* p, const Plan& pl, GemmDev& d) {
    d.M = p->M; d.N = p->N; d.K = p->K;
    d.A = p->A; d.lda = p->lda; d.B = p->B; d.ldb = p->ldb; d.C = p->C; d.ldc = p->ldc;
    d.epi = p->epilogue; d.bias = p->bias; d.pre = p->pre; d.residual = p->residual; d.ldr = p->ldr;
    d.aux = p->aux; d.rowscale = p->rowscale; d.rps = p->rows_per_scale > 0 ? p->rows_per_scale : 1;
    d.rowmap = p->rowmap;
    double th = (double)p->dropout_p * 4294967296.0;
    d.drop_thresh = th >= 4294967295.0 ? 4294967295u : (uint32_t)th;
    d.drop_scale = p->dropout_p < 1.0f ? 1.0f / (1.0f - p->dropout_p) : 0.0f;
    d.seed = p->seed; d.tag = p->tag;
    constexpr int E = TypeInfo<T>::E;
    const int bke = 128 / (int)sizeof(T);
    d.split_k = pl.split;
    int kps = ceil_div(ceil_div(p->K, bke), pl.split) * bke;
    d.k_per_split = pl.split > 1 ? kps : ((p->K + bke - 1) / bke) * bke;
    if (pl.split > 1) {
        // drop empty trailing splits
        d.split_k = ceil_div(p->K, kps);
        size_t need = (size_t)d.split_k * p->M * ((size_t)p->N + 1) * sizeof(float);
        MVLT_CHECK(p->workspace && p->workspace_bytes >= need, MVLT_ERR_ARG);
        if (d.split_k == 1) d.k_per_split = ((p->K + bke - 1) / bke) * bke;
    }
    d.ws = reinterpret_cast<float*>(p->workspace);
    d.a_colsum = p->a_colsum;
    d.pf = (p->prefetch && p->prefetch_bytes >= 128) ? reinterpret_cast<const char*>(p->prefetch) : nullptr;
    d.pf_lines = d.pf ? (long)(p->prefetch_bytes >> 7) : 0;
    d.m_dev = p->m_dev;
    d.a_kmajor = p->a_kmajor != 0;
    d.atomic_out = 0;
    d.ws_colsum = d.ws ? d.ws + (size_t)d.split_k * p->M * p->N : nullptr;
    d.a_vec = (p->lda % E == 0) && aligned16(p->A);
    d.b_vec = (p->ldb % E == 0) && aligned16(p->B);
    const int epi = p->epilogue;
    bool ev = (p->ldc % 4 == 0) && aligned16(p->C);
    if (epi & MVLT_EPI_RESIDUAL) ev = ev && (p->ldr % 4 == 0) && aligned16(p->residual);
    if (epi & MVLT_EPI_SAVE_PRE) ev = ev && aligned16(p->pre);
    if (epi & MVLT_EPI_MUL_GELU_GRAD) ev = ev && aligned16(p->aux);
    d.epi_vec = ev;
    // 16-byte row operands (tile_epilogue_wide): bf16 rows out, every row operand in 8-column chunks that are 16-byte aligned;
    // tiles of 64 / 128 columns only (fragment PAIRS)
    d.wide = sizeof(T) == 2 && ev && !(epi & (MVLT_EPI_OUT_F32 | MVLT_EPI_ACCUM)) && p->N % 8 == 0 && p->ldc % 8 == 0 && pl.split <= 1 &&
             pl.bn != 96 && (!(epi & MVLT_EPI_RESIDUAL) || p->ldr % 8 == 0) && (!(epi & MVLT_EPI_BIAS) || aligned16(p->bias));
    // tile order (tile_coords): the number of column groups that minimises what the eight L2s pull over the fabric,
    // M * xcs (rows of A, every group re-reads its row band) + 8 N / xcs (columns of B); MVLT_XCD_CS = 1 turns it off, 2 / 4 / 8 force it
    static const int xcs_env = [] { const char* e = getenv("MVLT_XCD_CS"); return e ? atoi(e) : 0; }();
    d.xcs = 1;
    {
        const int gx = ceil_div(p->N, pl.bn), gy = ceil_div(p->M, pl.bm);
        if (xcs_env > 1) { if (gx >= xcs_env) d.xcs = xcs_env; }
        else if (xcs_env == 0 && gx * gy >= 128) {
            long best = (long)p->M + 8L * p->N;
            for (int cs = 2; cs <= 8; cs *= 2)
                if (gx >= 2 * cs && gy >= 2 * (8 / cs)) {
                    const long c = (long)p->M * cs + 8L * p->N / cs;
                    if (c < best) { best = c; d.xcs = cs; }
                }
        }
    }
    return MVLT_OK;
}

template <typename T>
static int gemm_dispatch(const MvltGemm* p, hipStream_t s) {
    Plan pl = choose_plan(p);
    const bool skinny = is_skinny<T>(p) && (p->lda % TypeInfo<T>::E == 0) && (p->ldb % TypeInfo<T>::E == 0) &&
                        aligned16(p->A) && aligned16(p->B);
    if (is_skinny<T>(p)) pl.split = 1;          // (no workspace was requested for it)
    GemmDev d;
    { const int rc = fill_dev<T>(p, pl, d); if (rc != MVLT_OK) return rc; }
    if (skinny) {
        hipLaunchKernelGGL((gemm_skinny_kernel<T, false>), dim3(ceil_div(p->N, 16)), dim3(64 * SKINNY_WAVES), 0, s, d, ArgmaxOut{nullptr, nullptr});
        MVLT_LAUNCH_CHECK();
        if (p->event_after_main) (void)hipEventRecord(reinterpret_cast<hipEvent_t>(p->event_after_main), s);
        return MVLT_OK;
    }
    dim3 grid(ceil_div(p->N, pl.bn), ceil_div(p->M, pl.bm), d.split_k);
    const bool ak = p->a_kmajor != 0, bk = p->b_kmajor != 0;
    if constexpr (sizeof(T) == 2) {
        // weight-stationary row streaming for the HBM-bound Swin stage-0 / 1 products (100 k / 25 k rows, K, N <= 384)
        if (!ak && d.split_k <= 1 && p->M >= 16384 && p->N <= 768 && p->K <= 384) {
            const int rcs = mvlt_rowstream_try(&d, bk ? 1 : 0, s);
            if (rcs < 0) return MVLT_ERR_LAUNCH;
            if (rcs > 0) {
                if (p->event_after_main) (void)hipEventRecord(reinterpret_cast<hipEvent_t>(p->event_after_main), s);
                return MVLT_OK;
            }
        }
        // 8-wave ping-pong engine (gemm8.hip) for wide outputs: MVLT_G8 = 0 never / 1 wherever it is eligible
        const int g8m = g8_mode();
        if (g8m && !ak && d.split_k <= 1) {
            const int rc8 = mvlt_gemm8_try(&d, 1, 0, bk ? 1 : 0, g8m == 2, nullptr, nullptr, 0, s);
            if (rc8 < 0) return MVLT_ERR_LAUNCH;
            if (rc8 > 0) {
                if (p->event_after_main) (void)hipEventRecord(reinterpret_cast<hipEvent_t>(p->event_after_main), s);
                return MVLT_OK;
            }
        }
        // LDS-DMA loop: 64-row tiles only.  Standalone the LDS-DMA loop is 5-12 %
        // faster on 128x128 tiles and 18-24 % on 64x64; inside the training step the 128x128 form (64 KB of LDS, two
        // workgroups per CU) is SLOWER than the register-staged one (three per CU, shares the CU better with the
        // weight-gradient stream): 16.5 vs 16.1 ms per step.
        const bool glds_on = pl.bm == 64 || pl.bm == 160;          // LDS-DMA loop for the 64-row tiles (comment above) and the 160-row ones
        const int kspan = d.split_k > 1 ? d.k_per_split : p->K;
        const bool bkm_ok = !bk || ((pl.bn == 64 || pl.bn == 128) && p->N % 8 == 0 && p->N >= 8);
        if (glds_on && !ak && bkm_ok && p->K % 64 == 0 && kspan % 64 == 0 && d.a_vec && d.b_vec) {
            bool done = true;
#define GLDS_GO(BM_, BN_, BKM_) do { if (d.wide) hipLaunchKernelGGL((gemm_glds_kernel<BM_, BN_, BKM_, true>), grid, dim3(256), 0, s, d); \
                                     else hipLaunchKernelGGL((gemm_glds_kernel<BM_, BN_, BKM_, false>), grid, dim3(256), 0, s, d); } while (0)
            // 160-row tiles (72 KB of dynamic LDS, two workgroups per CU): wide epilogue only
#define GLDS_GO_DYN(BM_, BN_, BKM_) do { constexpr int sh_ = 2 * (BM_ + BN_) * 64 * 2; \
                static const bool ok_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds_kernel<BM_, BN_, BKM_, true>), \
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, sh_) == hipSuccess; \
                if (!ok_) return MVLT_ERR_LAUNCH; \
                hipLaunchKernelGGL((gemm_glds_kernel<BM_, BN_, BKM_, true>), grid, dim3(256), sh_, s, d); } while (0)
            // Three LDS stages (two K-tiles in flight per workgroup, counted vmcnt) where a launch has so few 64 x 64 tiles that a CU
            // holds one or two workgroups and every K-tile waits out a full L2 -> LDS latency: 1568 x 768 x 3072 26.8 -> 21.6 us.
            // With three workgroups per CU the loop is bound by the CU's L2 -> LDS bandwidth and the third stage only costs
            // (3150 x 768 x 3072: 32.3 vs 29.8 us; profiles/r5_glds_stages.txt).  MVLT_GLDS_STAGES=0 off, 2 = every 64 x 64 product.
            static const int st3 = [] { const char* e = getenv("MVLT_GLDS_STAGES"); return e ? atoi(e) : 1; }();
            const long tiles64 = (long)grid.x * grid.y;
            if (pl.bm == 160) {
                if (bk) GLDS_GO_DYN(160, 128, true); else GLDS_GO_DYN(160, 128, false);
            }
            else if (st3 && pl.bm == 64 && pl.bn == 64 && d.wide && kspan >= 768 && (st3 == 2 || (!bk && tiles64 <= 400))) {
                if (bk) hipLaunchKernelGGL((gemm_glds_kernel<64, 64, true, true, 3>), grid, dim3(256), 0, s, d);
                else hipLaunchKernelGGL((gemm_glds_kernel<64, 64, false, true, 3>), grid, dim3(256), 0, s, d);
            }
            else if (bk) {
                if (pl.bm == 128 && pl.bn == 128) GLDS_GO(128, 128, true);
                else if (pl.bm == 64 && pl.bn == 128) GLDS_GO(64, 128, true);
                else if (pl.bm == 64 && pl.bn == 64) GLDS_GO(64, 64, true);
                else done = false;
            }
            else if (pl.bm == 128 && pl.bn == 128) GLDS_GO(128, 128, false);
            else if (pl.bm == 128 && pl.bn == 96) GLDS_GO(128, 96, false);
            else if (pl.bm == 64 && pl.bn == 128) GLDS_GO(64, 128, false);
            else if (pl.bm == 64 && pl.bn == 96) GLDS_GO(64, 96, false);
            else if (pl.bm == 64 && pl.bn == 64) GLDS_GO(64, 64, false);
            else done = false;
#undef GLDS_GO
#undef GLDS_GO_DYN
            if (done) {
                MVLT_LAUNCH_CHECK();
                if (p->event_after_main) (void)hipEventRecord(reinterpret_cast<hipEvent_t>(p->event_after_main), s);
                if (d.split_k > 1) {
                    long total = (long)p->M * ((p->N + 3) / 4);
                    int blocks = (int)((total + 63) / 64);
                    if (blocks > 8192) blocks = 8192;
                    hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(blocks), dim3(256), 0, s, d);
                    MVLT_LAUNCH_CHECK();
                }
                return MVLT_OK;
            }
        }
    }
    if (pl.bm == 160) return MVLT_ERR_UNSUPPORTED;          // (tile160_ok and the conditions above disagree: never launch a mis-sized grid)
    if (pl.bm == 128 && pl.bn == 128) launch_layout<T, 128, 128>(d, ak, bk, grid, s);
    else if (pl.bm == 128 && pl.bn == 96) launch_layout<T, 128, 96>(d, ak, bk, grid, s);
    else if (pl.bm == 64 && pl.bn == 128) launch_layout<T, 64, 128>(d, ak, bk, grid, s);
    else if (pl.bm == 64 && pl.bn == 96) launch_layout<T, 64, 96>(d, ak, bk, grid, s);
    else launch_layout<T, 64, 64>(d, ak, bk, grid, s);
    MVLT_LAUNCH_CHECK();
    if (p->event_after_main) (void)hipEventRecord(reinterpret_cast<hipEvent_t>(p->event_after_main), s);
    if (d.split_k > 1) {
        long total = (long)p->M * ((p->N + 3) / 4);
        int blocks = (int)((total + 63) / 64);
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(blocks), dim3(256), 0, s, d);
        MVLT_LAUNCH_CHECK();
    }
    return MVLT_OK;
}

static int check_gemm_args(const MvltGemm* p) {
    MVLT_CHECK(p && p->A && p->B && p->C, MVLT_ERR_ARG);
    MVLT_CHECK(p->M > 0 && p->N > 0 && p->K > 0, MVLT_ERR_ARG);
    MVLT_CHECK(p->lda > 0 && p->ldb > 0 && p->ldc >= p->N, MVLT_ERR_ARG);
    MVLT_CHECK((long)p->M * p->N < (1L << 32) || !(p->epilogue & MVLT_EPI_DROPOUT), MVLT_ERR_ARG);
    const int e = p->epilogue;
    if (e & MVLT_EPI_BIAS) MVLT_CHECK(p->bias, MVLT_ERR_ARG);
    if (e & MVLT_EPI_SAVE_PRE) MVLT_CHECK(p->pre && (e & MVLT_EPI_GELU), MVLT_ERR_ARG);
    if (e & MVLT_EPI_RESIDUAL) MVLT_CHECK(p->residual && p->ldr >= p->N, MVLT_ERR_ARG);
    if (e & MVLT_EPI_ROWSCALE) MVLT_CHECK(p->rowscale, MVLT_ERR_ARG);
    if (e & MVLT_EPI_ROWMAP) MVLT_CHECK(p->rowmap, MVLT_ERR_ARG);
    if (e & MVLT_EPI_MUL_GELU_GRAD) MVLT_CHECK(p->aux, MVLT_ERR_ARG);
    if (e & MVLT_EPI_DROPOUT) MVLT_CHECK(p->dropout_p >= 0.f && p->dropout_p < 1.f, MVLT_ERR_ARG);
    if (p->a_colsum) MVLT_CHECK(p->a_kmajor, MVLT_ERR_ARG);
    return MVLT_OK;
}

// Grouped launch: weight gradients only (both operands k-major), 64-row tiles, no split-K.
template <typename T>
static int gemm_group_dispatch(const MvltGemm* items, int n, hipStream_t s) {
    GemmGroupDev g{};
    g.n = n;
    bool all128 = true, all96 = true;
    for (int i = 0; i < n; ++i) { all128 = all128 && items[i].N % 128 == 0; all96 = all96 && items[i].N % 96 == 0; }
    const int bn = all128 ? 128 : (all96 ? 96 : 0);
    MVLT_CHECK(bn != 0, MVLT_ERR_UNSUPPORTED);
    for (int i = 0; i < n; ++i) {
        const MvltGemm* p = items + i;
        MVLT_CHECK(p->a_kmajor && p->b_kmajor && p->dtype == items[0].dtype, MVLT_ERR_UNSUPPORTED);
        Plan pl{64, bn, 1};
        { const int rc = fill_dev<T>(p, pl, g.g[i]); if (rc != MVLT_OK) return rc; }
    }
    // 128-row tiles when they still give every CU a workgroup (half the LDS fragment traffic per MFMA)
    int bm = 64;
    if (bn == 128) {
        long t128 = 0;
        for (int i = 0; i < n; ++i) t128 += (long)ceil_div(items[i].M, 128) * ceil_div(items[i].N, 128);
        if (t128 >= 256) bm = 128;
        // MVLT_WGRAD_BM=64: 64 x 128 tiles for every group (162 instead of 243 VGPRs per wave: two such workgroups leave a third
        // of a SIMD's registers to the dgrad chain's kernels; A/B switch, profiles/r6_ln_bwd.md)
        static const int env_bm = [] { const char* e = getenv("MVLT_WGRAD_BM"); return e ? atoi(e) : 0; }();
        if (env_bm == 64) bm = 64;
    }
    long tiles = 0;
    int kmin = items[0].K;
    for (int i = 0; i < n; ++i) { tiles += (long)ceil_div(items[i].M, bm) * ceil_div(items[i].N, bn); kmin = items[i].K < kmin ? items[i].K : kmin; }
    // In-launch split-K: groups with fewer than 200 tiles (Swin stages 0/1: 21 / 72 tiles, 100k / 25k reduction rows) are
    // cut into k-slices: on the 8-wave engine below (f32 slabs, deterministic) when it takes the group, otherwise here, where the
    // slices meet in the zeroed f32 output through atomicAdd.  Groups with >= 200 tiles are never split: measured on the B=32 step it does not shorten
    // the stage-2 groups in situ and the zeroing launch + atomics cost 1 ms per step (17.1 vs 15.9 ms).
    const int bke = 128 / (int)sizeof(T);
    int split = 1;
    // the k-slices meet through f32 atomicAdd: the order of the 7-24 additions per element is not fixed, so the result is
    // not bit-reproducible run to run.  Never in the exact-f32 mode (the parity path), never under MVLT_DETERMINISTIC=1.
    static const bool deterministic = [] { const char* e = getenv("MVLT_DETERMINISTIC"); return e && e[0] == '1'; }();
    if (tiles < 200 && sizeof(T) == 2 && !deterministic) {
        split = (int)((512 + tiles - 1) / tiles);
        const int nkt = ceil_div(kmin, bke);
        if (split > nkt / 8) split = nkt / 8;
        if (split > 64) split = 64;
        if (split < 1) split = 1;
    }
    // 8-wave ping-pong engine (gemm8.hip) for the groups this kernel would cut into ATOMIC k-slices (few tiles, tens of
    // thousands of reduction rows: Swin stages 0 / 1): one persistent launch of 128 x 128 tiles, k-slices that meet through
    // f32 slabs (the last arriver of a tile sums them in slice order: deterministic, no float atomics), bias gradients as
    // dY^T . 1 on the matrix pipe -- 100 vs 137 us (stage 0) and 78 vs 91 us (stage 1) per group.  MVLT_G8=1 sends every
    // group there (tests / experiments: the BertLayer / stage-2 groups tie with this kernel), MVLT_G8=0 none.
    if constexpr (sizeof(T) == 2) {
        const int g8m = g8_mode();
        if (g8m && (g8m == 1 || split > 1 || tiles < 200)) {
            GemmDev tmp[GROUP_MAX];
            float* outs[GROUP_MAX];
            bool ok = true;
            for (int i = 0; i < n; ++i) {
                tmp[i] = g.g[i]; outs[i] = tmp[i].a_colsum; tmp[i].a_colsum = nullptr;
                ok = ok && tmp[i].epi == MVLT_EPI_OUT_F32;
            }
            if (ok) {
                const int rc8 = mvlt_gemm8_try(tmp, n, 1, 1, 0, outs, items[0].workspace, items[0].workspace_bytes, s);
                if (rc8 < 0) return MVLT_ERR_LAUNCH;
                if (rc8 > 0) return MVLT_OK;
            }
        }
    }
    g.split = split;
    for (int i = 0; i < n; ++i) {
        g.start[i] = i == 0 ? 0 : g.start[i];
        g.start[i + 1] = g.start[i] + ceil_div(items[i].M, bm) * ceil_div(items[i].N, bn) * split;
        if (split > 1) {
            GemmDev& d = g.g[i];
            d.split_k = split;
            d.k_per_split = ceil_div(ceil_div(items[i].K, bke), split) * bke;
            d.atomic_out = 1;
        }
    }
    if (split > 1) {
        MvltZeroItem z[2 * GROUP_MAX];
        int nz = 0;
        for (int i = 0; i < n; ++i) {
            MVLT_CHECK(items[i].ldc == items[i].N, MVLT_ERR_UNSUPPORTED);          // contiguous outputs (gradient arena views)
            z[nz++] = MvltZeroItem{reinterpret_cast<float*>(items[i].C), (int64_t)items[i].M * items[i].N};
            if (items[i].a_colsum) z[nz++] = MvltZeroItem{items[i].a_colsum, (int64_t)items[i].M};
        }
        const int rc = mvlt_zero_batch(z, nz, s);
        if (rc != MVLT_OK) return rc;
    }
    // at most 2 workgroups per CU (of the 3 that fit): the group runs on the side stream beside the dgrad
    // chain, which should keep a share of every CU (16.8 vs 17.1 ms/step uncapped; 1 / 3 per CU measured equal or worse)
    int total = g.start[n];
    // (MVLT_WGRAD_PER_CU=1|2|3: A/B switch; round 6 re-ran 1 against 2 with the low-footprint LayerNorm backward beside it)
    static const int per_cu = [] { const char* e = getenv("MVLT_WGRAD_PER_CU"); const int v = e ? atoi(e) : 2; return v >= 1 && v <= 3 ? v : 2; }();
    if (total > per_cu * 256) total = per_cu * 256;
    // bf16: two LDS stages + two register sets, one barrier per k-tile (the single-stage form measured 1.3 % slower in the step)
#define GROUP_LAUNCH(BM_, BN_, D_) hipLaunchKernelGGL((gemm_group_kernel<T, BM_, BN_, true, true, D_>), dim3(total), dim3(256), 0, s, g)
    if constexpr (sizeof(T) == 2) {
        // LDS-DMA form (round 6; MVLT_WGRAD_GLDS=0: the register-staged kernel): bf16, no k-slices, 128-wide column tiles, every
        // operand row 16-byte aligned in 8-element chunks, f32 output through the vector epilogue
        static const bool glds_wg = [] { const char* e = getenv("MVLT_WGRAD_GLDS"); return !e || atoi(e) != 0; }();
        // 128 x 128 tiles (BertLayer and stage-3 groups, two stages, two workgroups per CU): 62.2 against 77.1 us and 36.3 against 43.9 us
        // stand-alone.  64 x 128 tiles (stage 2: 216 tiles, ONE workgroup per CU): a lone workgroup needs the deeper pipeline -- two
        // stages 88 us, three stages (72 KB of dynamic LDS, two k-tiles in flight) 75 us, four 89 us, against 62 us for the
        // register-staged form stand-alone; INSIDE the step the order turns round, because the dgrad chain's kernels run beside
        // 128-register waves instead of 162-register ones: 11.45 / 11.45 / 11.36 ms per step with three stages against 11.55 / 11.50 /
        // 11.57 with the register-staged kernels (two stages 11.53 / 11.51 / 11.51; profiles/r6_wgrad_glds.md).
        // MVLT_WGRAD_GLDS: 0 = register-staged kernels, 1 = 128 x 128 groups only, 2 = 64 x 128 with two stages, 3 (default) = three.
        static const int glds_mode = [] { const char* e = getenv("MVLT_WGRAD_GLDS"); return e ? atoi(e) : 3; }();
        bool ok = glds_wg && split == 1 && bn == 128 && (bm == 128 || glds_mode >= 2);
        for (int i = 0; i < n && ok; ++i) {
            const GemmDev& d = g.g[i];
            ok = d.a_vec && d.b_vec && d.epi_vec && (d.M % 8 == 0) && (d.N % 8 == 0) && d.M >= 8 && d.N >= 8 &&
                 (d.epi & ~(MVLT_EPI_OUT_F32 | MVLT_EPI_ACCUM)) == 0 && (d.epi & MVLT_EPI_OUT_F32) && !d.atomic_out && d.split_k <= 1;
        }
        if (ok) {
            if (bm == 128) hipLaunchKernelGGL((gemm_group_glds_kernel<128, 128, 2>), dim3(total), dim3(256), 0, s, g);
            else if (glds_mode == 3) {          // 64 x 128 with three stages (72 KB of dynamic LDS: two k-tiles in flight for a lone workgroup)
                constexpr int sh3 = 3 * (64 + 128) * 64 * 2;
                static const bool ok3 = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_group_glds_kernel<64, 128, 3>),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, sh3) == hipSuccess;
                if (!ok3) return MVLT_ERR_LAUNCH;
                hipLaunchKernelGGL((gemm_group_glds_kernel<64, 128, 3>), dim3(total), dim3(256), sh3, s, g);
            }
            else hipLaunchKernelGGL((gemm_group_glds_kernel<64, 128, 2>), dim3(total), dim3(256), 0, s, g);
            MVLT_LAUNCH_CHECK();
            return MVLT_OK;
        }
        if (bn == 128 && bm == 128) GROUP_LAUNCH(128, 128, 2); else if (bn == 128) GROUP_LAUNCH(64, 128, 2); else GROUP_LAUNCH(64, 96, 2);
    } else {
        if (bn == 128 && bm == 128) GROUP_LAUNCH(128, 128, 0); else if (bn == 128) GROUP_LAUNCH(64, 128, 0); else GROUP_LAUNCH(64, 96, 0);
    }
#undef GROUP_LAUNCH
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}

// Workspace a grouped launch can use (items[0].workspace / workspace_bytes; 0 = none needed): room for the k-slice slabs of
// the 8-wave engine.  Without it the group still runs (unsliced, or on the 4-wave grouped kernel).
extern "C" size_t mvlt_gemm_group_workspace_bytes(const MvltGemm* items, int n) {
    if (!items || n < 1 || n > GROUP_MAX || items[0].dtype != MVLT_BF16) return 0;
    GemmDev tmp[GROUP_MAX];
    for (int i = 0; i < n; ++i) {
        if (check_gemm_args(items + i) != MVLT_OK || !items[i].a_kmajor || !items[i].b_kmajor) return 0;
        Plan pl{64, 128, 1};
        MvltGemm it = items[i];
        it.workspace = nullptr; it.workspace_bytes = 0;
        if (fill_dev<bf16_t>(&it, pl, tmp[i]) != MVLT_OK) return 0;
    }
    return mvlt_gemm8_group_workspace(tmp, n);
}

extern "C" int mvlt_gemm_group(const MvltGemm* items, int n, void* stream) {
    MVLT_CHECK(items && n >= 1 && n <= GROUP_MAX, MVLT_ERR_ARG);
    for (int i = 0; i < n; ++i) { const int rc = check_gemm_args(items + i); if (rc != MVLT_OK) return rc; }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (items[0].dtype == MVLT_F32) return gemm_group_dispatch<float>(items, n, s);
    if (items[0].dtype == MVLT_BF16) return gemm_group_dispatch<bf16_t>(items, n, s);
    return MVLT_ERR_UNSUPPORTED;
}

// the (max, index) partials per row and 16 columns: wide streaming form for big vocabularies, else the 16-column skinny kernel
template <typename T>
static void argmax_products(const MvltGemm* p, const GemmDev& d, float* part_val, int32_t* part_idx, int nblk, hipStream_t s) {
    const bool vec = (p->lda % TypeInfo<T>::E == 0) && (p->ldb % TypeInfo<T>::E == 0) && aligned16(p->A) && aligned16(p->B);
    if (vec && p->N >= 4096 && p->K % Mma<T>::KB == 0) {
        const dim3 grid(ceil_div(nblk, SKINNY_WAVES)), block(64 * SKINNY_WAVES);
        const ArgmaxOut am{part_val, part_idx};
        if (p->M <= 16) hipLaunchKernelGGL((gemm_argmax128_kernel<T, 1>), grid, block, 0, s, d, am, nblk);
        else if (p->M <= 32) hipLaunchKernelGGL((gemm_argmax128_kernel<T, 2>), grid, block, 0, s, d, am, nblk);
        else if (p->M <= 48) hipLaunchKernelGGL((gemm_argmax128_kernel<T, 3>), grid, block, 0, s, d, am, nblk);
        else hipLaunchKernelGGL((gemm_argmax128_kernel<T, 4>), grid, block, 0, s, d, am, nblk);
        return;
    }
    hipLaunchKernelGGL((gemm_skinny_kernel<T, true>), dim3(nblk), dim3(64 * SKINNY_WAVES), 0, s, d, ArgmaxOut{part_val, part_idx});
}

template <typename T>
static int gemm_argmax_dispatch(const MvltGemm* p, float* part_val, int32_t* part_idx, int64_t* out_idx, float* out_val,
                                hipStream_t s) {
    MVLT_CHECK(is_skinny<T>(p) && (p->lda % TypeInfo<T>::E == 0) && (p->ldb % TypeInfo<T>::E == 0) &&
               aligned16(p->A) && aligned16(p->B), MVLT_ERR_UNSUPPORTED);
    MVLT_CHECK((p->epilogue & ~(MVLT_EPI_BIAS)) == 0, MVLT_ERR_UNSUPPORTED);
    GemmDev d;
    Plan pl{64, 16, 1};
    { const int rc = fill_dev<T>(p, pl, d); if (rc != MVLT_OK) return rc; }
    const int nblk = ceil_div(p->N, 16);
    argmax_products<T>(p, d, part_val, part_idx, nblk, s);
    hipLaunchKernelGGL(argmax_parts_kernel, dim3(p->M), dim3(64), 0, s, part_val, part_idx, nblk, out_idx, out_val);
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}

extern "C" int mvlt_gemm_argmax(const MvltGemm* p, float* part_val, int32_t* part_idx, int64_t* out_idx, float* out_val,
                                void* stream) {
    MVLT_CHECK(p && p->A && p->B && part_val && part_idx && out_idx, MVLT_ERR_ARG);
    MVLT_CHECK(p->M > 0 && p->N > 0 && p->K > 0 && p->lda > 0 && p->ldb > 0, MVLT_ERR_ARG);
    if (p->epilogue & MVLT_EPI_BIAS) MVLT_CHECK(p->bias, MVLT_ERR_ARG);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (p->dtype == MVLT_F32) return gemm_argmax_dispatch<float>(p, part_val, part_idx, out_idx, out_val, s);
    if (p->dtype == MVLT_BF16) return gemm_argmax_dispatch<bf16_t>(p, part_val, part_idx, out_idx, out_val, s);
    return MVLT_ERR_UNSUPPORTED;
}

extern "C" int mvlt_gemm_argmax_greedy(const MvltGemm* p, float* part_val, int32_t* part_idx, const MvltGreedyState* g, void* stream) {
    MVLT_CHECK(p && p->A && p->B && part_val && part_idx && g, MVLT_ERR_ARG);
    MVLT_CHECK(p->M > 0 && p->M <= 64 && p->N > 0 && p->K > 0 && p->lda > 0 && p->ldb > 0, MVLT_ERR_ARG);
    MVLT_CHECK(!p->a_kmajor && !p->b_kmajor && (p->epilogue & ~(MVLT_EPI_BIAS)) == 0, MVLT_ERR_UNSUPPORTED);
    if (p->epilogue & MVLT_EPI_BIAS) MVLT_CHECK(p->bias, MVLT_ERR_ARG);
    MVLT_CHECK(g->col && g->ticket && g->ids && g->scores && g->new_ids && g->ld_ids > 0 && g->ld_scores > 0 && g->ld_new > 0, MVLT_ERR_ARG);
    if (g->has_eos) MVLT_CHECK(g->unfinished && g->alive, MVLT_ERR_ARG);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    GemmDev d;
    Plan pl{64, 16, 1};
    const int nblk = ceil_div(p->N, 16);
    // the same preconditions as mvlt_gemm_argmax (the skinny kernels walk whole k-blocks with 16-byte fragment loads: a K that is
    // not a multiple of the k-block would silently lose its tail, unaligned strides would fault) -- ADVICE r5
    MVLT_CHECK(g->ld_ids > 0 && g->ld_scores > 0 && g->ld_new > 0 && g->col != nullptr, MVLT_ERR_ARG);
    if (p->dtype == MVLT_BF16) {
        MVLT_CHECK(is_skinny<bf16_t>(p) && p->lda % TypeInfo<bf16_t>::E == 0 && p->ldb % TypeInfo<bf16_t>::E == 0 && aligned16(p->A) && aligned16(p->B), MVLT_ERR_UNSUPPORTED);
        const int rc = fill_dev<bf16_t>(p, pl, d); if (rc != MVLT_OK) return rc; argmax_products<bf16_t>(p, d, part_val, part_idx, nblk, s);
    } else if (p->dtype == MVLT_F32) {
        MVLT_CHECK(is_skinny<float>(p) && p->lda % TypeInfo<float>::E == 0 && p->ldb % TypeInfo<float>::E == 0 && aligned16(p->A) && aligned16(p->B), MVLT_ERR_UNSUPPORTED);
        const int rc = fill_dev<float>(p, pl, d); if (rc != MVLT_OK) return rc; argmax_products<float>(p, d, part_val, part_idx, nblk, s);
    } else return MVLT_ERR_UNSUPPORTED;
    GreedyState st{g->unfinished, g->eos_id, g->pad_id, g->has_eos, g->col, g->past, g->ids, g->ld_ids, g->scores, g->ld_scores, g->alive,
                   g->new_ids, g->ld_new, g->ticket};
    hipLaunchKernelGGL(greedy_pick_kernel, dim3(p->M), dim3(256), 0, s, part_val, part_idx, nblk, p->M, st);
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}

extern "C" int mvlt_gemm_skinny_accum(const MvltGemm* p, float* acc, int k_splits, void* stream) {
    MVLT_CHECK(p && p->A && p->B && acc && k_splits >= 1 && k_splits <= 64, MVLT_ERR_ARG);
    MVLT_CHECK(p->M > 0 && p->M <= 64 && p->N > 0 && p->K > 0 && !p->a_kmajor && !p->b_kmajor && p->epilogue == 0, MVLT_ERR_UNSUPPORTED);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    GemmDev d{};
    d.M = p->M; d.N = p->N; d.K = p->K; d.A = p->A; d.lda = p->lda; d.B = p->B; d.ldb = p->ldb;
    if (p->dtype == MVLT_BF16) {
        MVLT_CHECK(p->K % 32 == 0 && p->lda % 8 == 0 && p->ldb % 8 == 0 && aligned16(p->A) && aligned16(p->B), MVLT_ERR_UNSUPPORTED);
        hipLaunchKernelGGL((gemm_skinny_accum_kernel<bf16_t>), dim3(ceil_div(p->N, 16), k_splits), dim3(64 * SKINNY_WAVES), 0, s, d, acc);
    } else if (p->dtype == MVLT_F32) {
        MVLT_CHECK(p->K % 16 == 0 && p->lda % 4 == 0 && p->ldb % 4 == 0 && aligned16(p->A) && aligned16(p->B), MVLT_ERR_UNSUPPORTED);
        hipLaunchKernelGGL((gemm_skinny_accum_kernel<float>), dim3(ceil_div(p->N, 16), k_splits), dim3(64 * SKINNY_WAVES), 0, s, d, acc);
    } else return MVLT_ERR_UNSUPPORTED;
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}

extern "C" int mvlt_gemm(const MvltGemm* p, void* stream) {
    { const int rc = check_gemm_args(p); if (rc != MVLT_OK) return rc; }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (p->dtype == MVLT_F32) return gemm_dispatch<float>(p, s);
    if (p->dtype == MVLT_BF16) return gemm_dispatch<bf16_t>(p, s);
    return MVLT_ERR_UNSUPPORTED;
}
