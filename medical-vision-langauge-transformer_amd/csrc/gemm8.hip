// 8-wave "ping-pong" MFMA GEMM engine for gfx950 (bf16 in, f32 accumulate): the large products of the nn.Linear
// calls on the path -- forward x W^T and dgrad dy W of visual_feature_extractor.py:135-141,231,252 / HF
// modeling_bert.py:175-177,289,337,348 when the output is wide, and ALL weight gradients dW = dY^T X of one
// BertLayer / Swin block as one grouped launch.  Same argument block and epilogue semantics as gemm.hip (gemm_dev.h).
//
// One workgroup (512 threads = 8 waves, one per CU) owns a (128 MH) x (128 NH) output tile and walks a LIST of tiles
// (persistent: tile = blockIdx.x + i * gridDim.x over the concatenated tile lists of up to 8 products), so the first
// loads of tile i+1 are in flight while tile i is finished and stored.  Operands go global -> LDS by LDS-DMA
// (global_load_lds_dwordx4) in HALF-TILES of 128 operand rows x 64 k (16 KB, two 1-KB instructions per wave) through
// a ring of half-tile slots; per K-tile the order is [A0 | B0 | B1 | A1] (those that exist).
//   k-contiguous operand (x, dy, W in forward): image [128][64], 16-byte chunks XOR-swizzled by row & 7, ds_read_b128
//   k-major operand (W in dgrad, dY and X in wgrad): image [64 k][128], 32-byte units XOR-swizzled by kswz(k),
//       fragments transposed on the way out by ds_read_b64_tr_b16 -- no transposed copies in memory
//   (an LDS-DMA writes base + lane * 16, so both swizzles are applied to the per-lane SOURCE address)
// A wave (wr = wave / 4, wc = wave % 4) owns rows {wr 64 .. +64} of every A half and columns {wc 32 .. +32} of every
// B half: MH x NH quadrants of 64 x 32; one PHASE multiplies one quadrant over the 64-deep K-tile (16 MFMA 16x16x32).
// Fragments are read once per K-tile and kept in registers, so the half-tiles of a K-tile die in the order they
// were filled and the ring runs 1.25 - 2 K-tiles ahead with counted vmcnt waits (never 0 in the loop).
// The two wave groups (wr = 0 / 1; waves w and w + 4 share a SIMD) run ONE BARRIER apart: while one group's 16
// MFMAs occupy the SIMD's matrix pipe the other group issues its LDS reads and its LDS-DMA.
//
// Invariants (E_n = n-th barrier event; group 0 runs phase P between E_2P .. E_2P+2, group 1 between E_2P+1 .. E_2P+3):
//   RAW  in phase P every wave waits (counted vmcnt) for its own shares of every half-tile that phase P+1 reads
//        BEFORE phase P's first barrier, so every share has landed before anybody's phase P+1 reads.
//   WAR  a half-tile slot is re-filled no earlier than two phases after the last phase that read it: those reads have
//        completed in both groups before E_2L+3.
// Issues that would run past the end of the workgroup's tile list re-load the last tile into slots nobody reads any
// more (a few dozen KB of L2 hits per workgroup buy a loop body without a tail version).
#include "common.h"
#include "gemm_dev.h"
#include <cstdlib>

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

constexpr int HT_BYTES = 128 * 64 * 2;          // one half-tile: 128 operand rows x 64 bf16

// -DG8_TRACE (diagnostic build only): thread 0 of every workgroup stamps the 100 MHz real-time counter at the phase boundaries
// of its FIRST tile into a buffer set by mvlt_gemm8_trace_buffer (scripts/g8_trace.py reads it)
#ifdef G8_TRACE
__device__ long long* g_g8_trace = nullptr;
#define G8_STAMP(k) do { if (threadIdx.x == 0 && g_g8_trace) g_g8_trace[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define G8_STAMP(k) do { } while (0)
#endif

#define G8_FENCE() __builtin_amdgcn_sched_barrier(0)
#define G8_BARRIER() do { G8_FENCE(); __builtin_amdgcn_s_barrier(); G8_FENCE(); } while (0)
#define G8_VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

// LDS-DMA as inline asm: hipcc must NOT know that an LDS-DMA is in flight -- knowing it, it puts s_waitcnt vmcnt(0) in
// front of every ds_read_b64_tr_b16 (the intrinsic carries no alias information), which drains the ring in every phase.
// The counted vmcnt waits of the loop are all written by hand anyway.  M0 (the LDS destination) is compiler-reserved:
// saved and restored inside the statement.  lds_dst must be wave-uniform.
MVLT_DEV void g8_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

MVLT_DEV int g8_kswz(int k) { return (k & 3) | ((k >> 1) & 4); }          // == kswz<128> of gemm.hip

// fragment (16 operand rows x 32 k) of a k-contiguous half-tile image [128][64]
MVLT_DEV bf16x8 g8_frag_rm(const char* half, int row0, int kb, int lane) {
    const int row = row0 + (lane & 15);
    const int ch = (kb * 4 + (lane >> 4)) ^ (lane & 7);          // (row & 7) == (lane & 7): row0 is a multiple of 8
    return *reinterpret_cast<const bf16x8*>(half + row * 128 + ch * 16);
}
// the same fragment of a k-major half-tile image [64 k][128 rows]: two transposing reads (4 k x 16 rows each)
MVLT_DEV bf16x8 g8_frag_km(const char* half, int row0, int kb, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
    const int k = kb * 32 + 8 * g + q, c = row0 >> 4;
    const char* p0 = half + k * 256 + ((c ^ g8_kswz(k)) << 5) + 8 * pp;
    const char* p1 = half + (k + 4) * 256 + ((c ^ g8_kswz(k + 4)) << 5) + 8 * pp;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p0);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p1);
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
template <bool KM> MVLT_DEV bf16x8 g8_frag(const char* half, int row0, int kb, int lane) {
    if constexpr (KM) return g8_frag_km(half, row0, kb, lane); else return g8_frag_rm(half, row0, kb, lane);
}

// Epilogue of one tile with the flag set known at compile time (the generic epilogue4 of gemm_dev.h tests every flag at
// run time for each of the 32 accumulator fragments of a wave: ~200 KB of straight-line code, ~10 us per launch in
// instruction fetch alone).  Same order of operations as epilogue4; a lane owns 4 consecutive columns of a fragment.
// vmcnt completes IN ORDER, loads and stores alike: a load issued behind a store cannot be consumed before that store
// has completed (~1 us under load), so an epilogue that loads bias / residual / aux fragment by fragment between its
// stores pays a store latency per fragment (+23 us on the BERT FFN-in product).  Here every load of a row block is
// issued BEFORE the stores of the previous row block: the bias values (4 column groups per lane) once per tile, the
// per-row values and the residual / aux fragments one row block ahead.
// WIDE (g8_tile_epilogue_wide): the 8 values are the lane's eight consecutive columns of the chunk layout
// (n_base + 128 c + 16 (g & 1) + 8 (g >> 1), g = lane / 16), [0] the lower four, [1] the upper four.
template <int NH, int EPI, bool WIDE>
MVLT_DEV void g8_load_bias(const GemmDev& p, const int n_base, const int lane, f32x4 (&bias_v)[NH][2]) {
    if constexpr ((EPI & MVLT_EPI_BIAS) != 0) {
        const int g = lane >> 4;
#pragma unroll
        for (int c = 0; c < NH; ++c)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = WIDE ? min(n_base + c * 128 + 16 * (g & 1) + 8 * (g >> 1), p.N - 8) + 4 * j          // N % 8 == 0 (GemmDev.wide)
                                   : min(n_base + c * 128 + j * 16 + 4 * g, p.N - 4);                               // N % 4 == 0 (launcher)
                bias_v[c][j] = *reinterpret_cast<const f32x4*>(p.bias + n);
            }
    }
}

// No divergent control flow around any LOAD (out-of-range rows / columns are clamped to valid ones, only the stores are
// predicated): hipcc's wait insertion then counts exactly; a load whose first use sits in a divergent block gets a
// vmcnt(0) in EVERY such block, i.e. a wait for all older stores per fragment.
template <int MH, int NH, int EPI>
MVLT_DEV void g8_tile_epilogue(const GemmDev& p, const int m_base, const int n_base, const int lane, f32x4 (&acc)[MH][NH][4][2],
                               const f32x4 (&bias_v)[NH][2]) {
    constexpr bool HAS_ROWLOAD = (EPI & (MVLT_EPI_ROWMAP | MVLT_EPI_ROWSCALE)) != 0;
    constexpr bool HAS_FRAGLOAD = (EPI & (MVLT_EPI_RESIDUAL | MVLT_EPI_MUL_GELU_GRAD)) != 0;
    constexpr int NRB = MH * 4;                                           // row blocks of 16 rows
    const int mr = lane & 15, nq = 4 * (lane >> 4);
    auto col_of = [&](int c, int j) { return n_base + c * 128 + j * 16 + nq; };
    auto row_of = [&](int rb) { return m_base + (rb >> 2) * 128 + (rb & 3) * 16 + mr; };
    f32x4 bv[NH][2];
    if constexpr ((EPI & MVLT_EPI_BIAS) != 0) {
        // an opaque use in straight-line code: hipcc waits for the bias loads HERE, once (it would otherwise sink their
        // first use into the predicated store blocks below and wait vmcnt(0), i.e. for every older store, in each one)
#pragma unroll
        for (int c = 0; c < NH; ++c)
#pragma unroll
            for (int j = 0; j < 2; ++j) { bv[c][j] = bias_v[c][j]; asm volatile("" : "+v"(bv[c][j])); }
    }
    struct RowPre { int mo; float sc; bf16x4 res[NH][2]; bf16x4 aux[NH][2]; };
    auto preload = [&](int rb, RowPre& r) {
        const int m = min(row_of(rb), p.M - 1);
        r.mo = m; r.sc = 1.0f;
        if constexpr ((EPI & MVLT_EPI_ROWMAP) != 0) r.mo = p.rowmap[m];
        if constexpr ((EPI & MVLT_EPI_ROWSCALE) != 0) r.sc = p.rowscale[r.mo / p.rps];
        if constexpr (HAS_FRAGLOAD) {
#pragma unroll
            for (int c = 0; c < NH; ++c)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int n = min(col_of(c, j), p.N - 4);
                    if constexpr ((EPI & MVLT_EPI_RESIDUAL) != 0)
                        r.res[c][j] = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(p.residual) + (long)r.mo * p.ldr + n);
                    if constexpr ((EPI & MVLT_EPI_MUL_GELU_GRAD) != 0)
                        r.aux[c][j] = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(p.aux) + (long)r.mo * p.ldc + n);
                }
        }
    };
    // loads run TWO row blocks ahead of the stores; `pin` is an opaque use in straight-line code, so hipcc waits for a
    // row block's loads there, once, with the younger loads (and most of the previous row block's stores) still in flight
    auto pin = [&](RowPre& r) {
        if constexpr (HAS_FRAGLOAD) {
#pragma unroll
            for (int c = 0; c < NH; ++c)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr ((EPI & MVLT_EPI_RESIDUAL) != 0) asm volatile("" : "+v"(r.res[c][j]));
                    if constexpr ((EPI & MVLT_EPI_MUL_GELU_GRAD) != 0) asm volatile("" : "+v"(r.aux[c][j]));
                }
        }
        if constexpr ((EPI & MVLT_EPI_ROWSCALE) != 0) asm volatile("" : "+v"(r.sc));
    };
    RowPre cur, nxt, nx2;
    if constexpr (HAS_ROWLOAD || HAS_FRAGLOAD) { preload(0, nxt); preload(1, nx2); }
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
        const int m = row_of(rb);
        if constexpr (HAS_ROWLOAD || HAS_FRAGLOAD) {
            cur = nxt; nxt = nx2;
            if (rb + 2 < NRB) preload(rb + 2, nx2);
            pin(cur);
        } else { cur.mo = m; cur.sc = 1.0f; }
#pragma unroll
        for (int c = 0; c < NH; ++c)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = col_of(c, j);
                const bool live = m < p.M && n < p.N;
                f32x4 v = acc[rb >> 2][c][rb & 3][j];
                if constexpr ((EPI & MVLT_EPI_BIAS) != 0) v += bv[c][j];
                const long co = (long)cur.mo * p.ldc + n;
                if constexpr ((EPI & MVLT_EPI_GELU) != 0) {
                    if constexpr ((EPI & MVLT_EPI_SAVE_PRE) != 0) { if (live) store4f(reinterpret_cast<bf16_t*>(p.pre) + co, v); }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
                }
                if constexpr ((EPI & MVLT_EPI_DROPOUT) != 0) {
                    const uint32_t base = (uint32_t)m * (uint32_t)p.N + (uint32_t)n;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = rng_keep(p.seed, p.tag, base + e, p.drop_thresh) ? v[e] * p.drop_scale : 0.0f;
                }
                if constexpr ((EPI & MVLT_EPI_ROWSCALE) != 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= cur.sc;
                }
                if constexpr ((EPI & MVLT_EPI_MUL_GELU_GRAD) != 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= gelu_grad_f((float)cur.aux[c][j][e]);
                }
                if constexpr ((EPI & MVLT_EPI_RESIDUAL) != 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)cur.res[c][j][e];
                }
                if (live) {
                    if constexpr ((EPI & MVLT_EPI_OUT_F32) != 0) store4f(reinterpret_cast<float*>(p.C) + co, v);
                    else store4f(reinterpret_cast<bf16_t*>(p.C) + co, v);
                }
            }
    }
}

// The same epilogue with 16 bytes per lane on every row operand (gemm_dev.h tile_epilogue_wide has the derivation): the two
// 16-column fragments (j = 0, 1) of a quadrant's column block trade halves through v_permlane16_swap, after which a lane
// holds EIGHT consecutive columns of one row -- 16 instead of 32 row stores per lane and tile (32 instead of 64 with a saved
// pre-activation), every row load 16 bytes.  The store tail of the 256 x 256 tile is bound by the ISSUE of these
// instructions, not by their bytes (cdna_hip_programming.md T21).  Chosen per launch (GemmDev.wide).
template <int MH, int NH, int EPI>
MVLT_DEV void g8_tile_epilogue_wide(const GemmDev& p, const int m_base, const int n_base, const int lane, f32x4 (&acc)[MH][NH][4][2],
                                    const f32x4 (&bias_v)[NH][2]) {
    static_assert((EPI & (MVLT_EPI_OUT_F32 | MVLT_EPI_DROPOUT)) == 0, "bf16 rows out");
    constexpr bool HAS_ROWLOAD = (EPI & (MVLT_EPI_ROWMAP | MVLT_EPI_ROWSCALE)) != 0;
    constexpr bool HAS_FRAGLOAD = (EPI & (MVLT_EPI_RESIDUAL | MVLT_EPI_MUL_GELU_GRAD)) != 0;
    constexpr int NRB = MH * 4;
    const int mr = lane & 15, g = lane >> 4;
    const int cofs = 16 * (g & 1) + 8 * (g >> 1);
    auto col_of = [&](int c) { return n_base + c * 128 + cofs; };
    auto row_of = [&](int rb) { return m_base + (rb >> 2) * 128 + (rb & 3) * 16 + mr; };
    f32x4 bv[NH][2];
    if constexpr ((EPI & MVLT_EPI_BIAS) != 0) {
#pragma unroll
        for (int c = 0; c < NH; ++c)
#pragma unroll
            for (int j = 0; j < 2; ++j) { bv[c][j] = bias_v[c][j]; asm volatile("" : "+v"(bv[c][j])); }
    }
    struct RowPre { int mo; float sc; u32x4 res[NH]; u32x4 aux[NH]; };
    auto preload = [&](int rb, RowPre& r) {
        const int m = min(row_of(rb), p.M - 1);
        r.mo = m; r.sc = 1.0f;
        if constexpr ((EPI & MVLT_EPI_ROWMAP) != 0) r.mo = p.rowmap[m];
        if constexpr ((EPI & MVLT_EPI_ROWSCALE) != 0) r.sc = p.rowscale[r.mo / p.rps];
        if constexpr (HAS_FRAGLOAD) {
#pragma unroll
            for (int c = 0; c < NH; ++c) {
                const int n = min(col_of(c), p.N - 8);
                if constexpr ((EPI & MVLT_EPI_RESIDUAL) != 0)
                    r.res[c] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(p.residual) + (long)r.mo * p.ldr + n);
                if constexpr ((EPI & MVLT_EPI_MUL_GELU_GRAD) != 0)
                    r.aux[c] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(p.aux) + (long)r.mo * p.ldc + n);
            }
        }
    };
    auto pin = [&](RowPre& r) {
        if constexpr (HAS_FRAGLOAD) {
#pragma unroll
            for (int c = 0; c < NH; ++c) {
                if constexpr ((EPI & MVLT_EPI_RESIDUAL) != 0) asm volatile("" : "+v"(r.res[c]));
                if constexpr ((EPI & MVLT_EPI_MUL_GELU_GRAD) != 0) asm volatile("" : "+v"(r.aux[c]));
            }
        }
        if constexpr ((EPI & MVLT_EPI_ROWSCALE) != 0) asm volatile("" : "+v"(r.sc));
    };
    RowPre cur, nxt, nx2;
    if constexpr (HAS_ROWLOAD || HAS_FRAGLOAD) { preload(0, nxt); preload(1, nx2); }
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
        const int m = row_of(rb);
        if constexpr (HAS_ROWLOAD || HAS_FRAGLOAD) {
            cur = nxt; nxt = nx2;
            if (rb + 2 < NRB) preload(rb + 2, nx2);
            pin(cur);
        } else { cur.mo = m; cur.sc = 1.0f; }
#pragma unroll
        for (int c = 0; c < NH; ++c) {
            const int n = col_of(c);
            const bool live = m < p.M && n < p.N;                          // (N % 8 == 0: a chunk is inside or outside as a whole)
            f32x4 lo = acc[rb >> 2][c][rb & 3][0], hi = acc[rb >> 2][c][rb & 3][1];
            swap16(lo, hi);
            if constexpr ((EPI & MVLT_EPI_BIAS) != 0) { lo += bv[c][0]; hi += bv[c][1]; }
            const long co = (long)cur.mo * p.ldc + n;
            if constexpr ((EPI & MVLT_EPI_GELU) != 0) {
                if constexpr ((EPI & MVLT_EPI_SAVE_PRE) != 0) { if (live) *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.pre) + co) = pack8(lo, hi); }
#pragma unroll
                for (int e = 0; e < 4; ++e) { lo[e] = gelu_f(lo[e]); hi[e] = gelu_f(hi[e]); }
            }
            if constexpr ((EPI & MVLT_EPI_ROWSCALE) != 0) { lo *= cur.sc; hi *= cur.sc; }
            if constexpr ((EPI & MVLT_EPI_MUL_GELU_GRAD) != 0) {
                f32x4 alo, ahi;
                unpack8(cur.aux[c], alo, ahi);
#pragma unroll
                for (int e = 0; e < 4; ++e) { lo[e] *= gelu_grad_f(alo[e]); hi[e] *= gelu_grad_f(ahi[e]); }
            }
            if constexpr ((EPI & MVLT_EPI_RESIDUAL) != 0) {
                f32x4 rlo, rhi;
                unpack8(cur.res[c], rlo, rhi);
                lo += rlo; hi += rhi;
            }
            if (live) *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.C) + co) = pack8(lo, hi);
        }
    }
}

constexpr int G8_GROUP_MAX = 8;
struct G8Group {
    int n; GemmDev g[G8_GROUP_MAX]; const void* zero_page;
    // split-K (weight gradients with too few output tiles for 256 CUs): every tile is cut into `split` k-slices = units;
    // a unit writes its partial tile to slabs[tile][slice] (f32, fragment order), draws a ticket from counters[tile], and
    // the LAST arriver sums the slices in slice order (deterministic, no float atomics) and runs the epilogue.
    int split; float* slabs; float* cs_slabs; int* counters;
    float* colsum[G8_GROUP_MAX];          // weight gradients: bias gradient of product i (column sums of its k-major A), or null
};

template <int MH, int NH> struct G8Cfg {
    static constexpr int HPK = MH + NH;                                 // half-tiles per K-tile
    static constexpr int RING_KT = HPK == 4 ? 2 : (HPK == 3 ? 3 : 4);   // K-tiles the LDS ring holds
    static constexpr int LDS = RING_KT * HPK * HT_BYTES;
};

// The tile whose half-tiles are being ISSUED (runs ahead of the tile being multiplied).
template <int MH, int NH> struct G8Issue {
    const bf16_t* pa[MH][2];    // [half][instruction]: this lane's source of the A half at k = 0 of the product
    const bf16_t* pb[NH][2];
    long ka, kb;                // element step per k (1 for a k-contiguous operand, ld for a k-major one)
    int K;                      // reduction length of the issue unit's product
    int kt, kt_end;             // K-tile the next A0 belongs to / end of the issue unit's K-tile range
    int ord;                    // ordinal of the issue unit in this workgroup's list
};

template <int MH, int NH, bool AKM, bool BKM, int EPI, bool WIDE = false>
__global__ __launch_bounds__(512, 1) void gemm8_kernel(const G8Group gp) {
    using Cfg = G8Cfg<MH, NH>;
    constexpr int BM = 128 * MH, BN = 128 * NH, HPK = Cfg::HPK, RING_KT = Cfg::RING_KT;
    extern __shared__ __attribute__((aligned(1024))) char smem[];          // ring of RING_KT * HPK half-tile slots
    G8_STAMP(0);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wr = wave >> 2, wc = wave & 3;
    const int G = gridDim.x, b = blockIdx.x;

    // ---- tile list: the products' tile lists concatenated (effective sizes: ragged batches read their row count here)
    // m_dev is read ONCE per product here (a global load inside the K loop would wait vmcnt(0) and drain the ring); the
    // effective sizes live in LDS behind the ring (a private array indexed by the product would go to scratch)
    int* eff_lds = reinterpret_cast<int*>(smem + Cfg::LDS);               // [G8_GROUP_MAX]: M (k-contiguous A) or K (weight gradients)
    int start[G8_GROUP_MAX + 1];
    start[0] = 0;
#pragma unroll
    for (int i = 0; i < G8_GROUP_MAX; ++i) {
        int t = 0;
        if (i < gp.n) {
            const GemmDev q = effective<AKM>(gp.g[i]);
            if (threadIdx.x == 0) eff_lds[i] = AKM ? q.K : q.M;
            t = ((q.M + BM - 1) / BM) * ((q.N + BN - 1) / BN);
        }
        start[i + 1] = start[i] + t;
    }
    __syncthreads();
    auto product = [&](int item) {
        GemmDev q = gp.g[item];
        const int e = __builtin_amdgcn_readfirstlane(eff_lds[item]);
        if (AKM) q.K = e; else q.M = e;
        return q;
    };
    const int ntiles = start[G8_GROUP_MAX];
    const int S = gp.split > 1 ? gp.split : 1;
    const int nunits = ntiles * S;                                        // unit = (tile, k-slice)
    if (b >= nunits) return;
    const int nt_wg = (nunits - b + G - 1) / G;

    // unit ordinal -> (product, by, bx, tile, slice, K-tile range); past the end of the list: the last unit (only re-loaded)
    struct Unit { int item, by, bx, tile, slice, kt0, kt1; };
    auto locate = [&](int ord) {
        Unit u;
        const int un = xcd_remap(min(b + ord * G, nunits - 1), nunits);   // slices of a tile are neighbours: same XCD chunk
        const int t = un / S;
        u.tile = t; u.slice = un - t * S;
        int item = 0;
#pragma unroll
        for (int i = 1; i < G8_GROUP_MAX; ++i) if (i < gp.n && t >= start[i]) item = i;
        const int gx = (gp.g[item].N + BN - 1) / BN;
        const int local = t - start[item];
        const int by = local / gx;
        u.item = __builtin_amdgcn_readfirstlane(item); u.by = __builtin_amdgcn_readfirstlane(by);
        u.bx = __builtin_amdgcn_readfirstlane(local - by * gx);
        const int kdim = AKM ? __builtin_amdgcn_readfirstlane(eff_lds[u.item]) : gp.g[u.item].K;
        const int nk = (kdim + 63) >> 6;
        const int per = (nk + S - 1) / S;
        u.kt0 = min(u.slice * per, nk); u.kt1 = min(u.kt0 + per, nk);
        return u;
    };

    G8Issue<MH, NH> is;
    const bf16_t* zero_page = reinterpret_cast<const bf16_t*>(gp.zero_page);
    auto set_tile = [&](int ord) {
        // units with an empty K-tile range (fewer K-tiles than slices) multiply nothing and are skipped by the issue
        // stream as they are by the K loop; past the end of the list any unit serves (its loads are never read)
        Unit u = locate(ord);
        while (u.kt0 >= u.kt1 && ord < nt_wg) { ++ord; u = locate(ord); }
        if (ord >= nt_wg) { u.kt0 = 0; u.kt1 = 1; }
        const int by = u.by, bx = u.bx;
        const GemmDev q = product(u.item);
        const bf16_t* A = reinterpret_cast<const bf16_t*>(q.A);
        const bf16_t* B = reinterpret_cast<const bf16_t*>(q.B);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int h = 0; h < MH; ++h) {
                if constexpr (AKM) {          // instruction = 4 k-rows x 256 B; lane = (k-row, 16-byte position x)
                    const int k = (wave * 2 + j) * 4 + (lane >> 4), x = lane & 15;
                    const int chunk = (((x >> 1) ^ g8_kswz(k)) << 1) | (x & 1);
                    const int col = min(by * BM + h * 128 + chunk * 8, max(q.M - 8, 0));
                    is.pa[h][j] = A + (long)k * q.lda + col;
                } else {                      // instruction = 8 rows x 128 B; lane = (row, chunk')
                    const int rin = lane >> 3, chs = (lane & 7) ^ rin;
                    const int row = min(by * BM + h * 128 + (wave * 2 + j) * 8 + rin, q.M - 1);
                    is.pa[h][j] = A + (long)row * q.lda + chs * 8;
                }
            }
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                if constexpr (BKM) {
                    const int k = (wave * 2 + j) * 4 + (lane >> 4), x = lane & 15;
                    const int chunk = (((x >> 1) ^ g8_kswz(k)) << 1) | (x & 1);
                    const int col = min(bx * BN + h * 128 + chunk * 8, max(q.N - 8, 0));
                    is.pb[h][j] = B + (long)k * q.ldb + col;
                } else {
                    const int rin = lane >> 3, chs = (lane & 7) ^ rin;
                    const int row = min(bx * BN + h * 128 + (wave * 2 + j) * 8 + rin, q.N - 1);
                    is.pb[h][j] = B + (long)row * q.ldb + chs * 8;
                }
            }
        }
        is.ka = AKM ? q.lda : 1; is.kb = BKM ? q.ldb : 1;
        is.K = q.K; is.kt = u.kt0; is.kt_end = u.kt1;
        is.ord = ord;
    };
    // half-tile `typ` (0 A0, 1 B0, 2 B1, 3 A1) of K-tile `kt` of the issue tile into ring buffer `d`
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    auto issue = [&](int typ, int kt, int d) {
        const int slot_in_kt = typ == 0 ? 0 : (typ == 1 ? 1 : (typ == 2 ? 2 : HPK - 1));
        const unsigned slot = lds_base + (d * HPK + slot_in_kt) * HT_BYTES + wave * 2048;
        const bool is_a = typ == 0 || typ == 3;
        const int h = (typ == 3 || typ == 2) ? 1 : 0;
        // weight gradients, last K-tile of a ragged reduction: rows beyond K read zeros (A) / the last valid row (B: finite,
        // x * 0 = 0) whatever the buffers hold there.  Uniform branch: every other K-tile takes the plain path.
        const bool tail = AKM && BKM && (kt + 1) * 64 > is.K;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bf16_t* src;
            if (is_a) src = is.pa[h < MH ? h : 0][j] + (long)kt * 64 * is.ka;
            else src = is.pb[h < NH ? h : 0][j] + (long)kt * 64 * is.kb;
            if constexpr (AKM && BKM) {
                if (tail) {
                    const int k = kt * 64 + (wave * 2 + j) * 4 + (lane >> 4);
                    if (is_a) src = k < is.K ? src : zero_page + (lane & 15) * 8;
                    else src = k < is.K ? src : src - (long)(k - max(is.K - 1, 0)) * is.kb;
                }
            }
#ifndef G8_NO_GLDS          /* ablation build: no LDS-DMA (stale LDS is multiplied; outputs wrong) */
            g8_glds16(src, slot + j * 1024);
#else
            asm volatile("" :: "v"(src), "s"(slot));
#endif
        }
    };
    auto advance = [&]() { if (++is.kt >= is.kt_end) set_tile(is.ord + 1); };

    // ---- prologue
    G8_STAMP(1);
    set_tile(0);
    if constexpr (HPK == 4) {                  // the whole K-tile 0 and A0 of K-tile 1; 3 half-tiles stay in flight
        issue(0, is.kt, 0); issue(1, is.kt, 0); issue(2, is.kt, 0); issue(3, is.kt, 0);
        advance();
        issue(0, is.kt, 1);
        G8_VMCNT(6);
    } else if constexpr (HPK == 3) {           // K-tiles 0 and 1; A0, B0 of K-tile 0 must have landed
        issue(0, is.kt, 0); issue(1, is.kt, 0); issue(MH == 2 ? 3 : 2, is.kt, 0);
        advance();
        issue(0, is.kt, 1); issue(1, is.kt, 1); issue(MH == 2 ? 3 : 2, is.kt, 1);
        G8_VMCNT(8);
    } else {
        issue(0, is.kt, 0); issue(1, is.kt, 0);
        advance();
        issue(0, is.kt, 1); issue(1, is.kt, 1);
        G8_VMCNT(4);
    }
    G8_STAMP(2);
    G8_BARRIER();
    G8_STAMP(3);
    if (wr == 1) G8_BARRIER();                                            // group 1 runs one barrier behind

    int g = 0;                                                            // flat K-tile index over the tile list
    const bf16x8 ones = {(bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f};
    for (int ord = 0; ord < nt_wg; ++ord) {
        const Unit un = locate(ord);
        const int by = un.by, bx = un.bx;
        const GemmDev p = product(un.item);
        // bias values of this lane's 2 NH column groups: loaded now, used after the K loop (a load issued in the epilogue
        // would queue behind the ring's in-flight LDS-DMA: vmcnt completes in order)
        f32x4 bias_v[NH][2];
        g8_load_bias<NH, EPI, WIDE>(p, bx * BN + wc * 32, lane, bias_v);
        f32x4 acc[MH][NH][4][2];
#pragma unroll
        for (int a = 0; a < MH; ++a)
#pragma unroll
            for (int c = 0; c < NH; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[a][c][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // weight gradients: the bias gradient db[m] = sum_k dY[k][m] rides along as dY^T . 1 -- wave wc multiplies the A
        // fragment of its row block wc with a fragment of ones (2 extra MFMAs per A half and K-tile, +3-6 %)
        f32x4 cs[MH];
#pragma unroll
        for (int a = 0; a < MH; ++a) cs[a] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int kt = un.kt0; kt < un.kt1; ++kt, ++g) {
            const int d = g % RING_KT;
            const char* base = smem + d * HPK * HT_BYTES;
            bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
#define G8_READ_A(HALF_SLOT) _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) \
                fa[i][kb] = g8_frag<AKM>(base + (HALF_SLOT) * HT_BYTES, wr * 64 + i * 16, kb, lane)
#define G8_READ_B(DST, HALF_SLOT) _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) \
                DST[j][kb] = g8_frag<BKM>(base + (HALF_SLOT) * HT_BYTES, wc * 32 + j * 16, kb, lane)
#ifdef G8_NO_MMA          /* ablation build: LDS reads kept alive, no MFMA (outputs wrong) */
#define G8_MMA(C_, FB_) do { \
                _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) { \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(fa[i][kb])); \
                _Pragma("unroll") for (int j = 0; j < 2; ++j) asm volatile("" :: "v"(FB_[j][kb])); } } while (0)
#else
#define G8_MMA(C_, FB_) do { \
                __builtin_amdgcn_s_setprio(1); \
                _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) \
                _Pragma("unroll") for (int j = 0; j < 2; ++j) Mma<bf16_t>::mma(C_[i][j], FB_[j][kb], fa[i][kb]); \
                __builtin_amdgcn_s_setprio(0); } while (0)
#endif
#define G8_CS(A_) do { if constexpr (AKM && BKM) { \
                _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) { \
                    if (wc == 0) Mma<bf16_t>::mma(cs[A_], ones, fa[0][kb]); else if (wc == 1) Mma<bf16_t>::mma(cs[A_], ones, fa[1][kb]); \
                    else if (wc == 2) Mma<bf16_t>::mma(cs[A_], ones, fa[2][kb]); else Mma<bf16_t>::mma(cs[A_], ones, fa[3][kb]); } } } while (0)
            if constexpr (MH == 2 && NH == 2) {
                // slots A0 B0 B1 A1; phases (0,0) (0,1) (1,1) (1,0); one half-tile issued per phase, 5 half-tiles ahead
                G8_READ_B(fb0, 1); G8_READ_A(0);
                G8_FENCE(); issue(1, is.kt, (g + 1) % RING_KT); G8_VMCNT(6);
                G8_BARRIER(); G8_MMA(acc[0][0], fb0); G8_CS(0); G8_BARRIER();
                G8_READ_B(fb1, 2);
                G8_FENCE(); issue(2, is.kt, (g + 1) % RING_KT); G8_VMCNT(6);
                G8_BARRIER(); G8_MMA(acc[0][1], fb1); G8_BARRIER();
                G8_READ_A(3);
                G8_FENCE(); issue(3, is.kt, (g + 1) % RING_KT); G8_VMCNT(6);
                G8_BARRIER(); G8_MMA(acc[1][1], fb1); G8_CS(MH - 1); G8_BARRIER();
                advance(); issue(0, is.kt, d); G8_VMCNT(6);
                G8_BARRIER(); G8_MMA(acc[1][0], fb0); G8_BARRIER();
            } else if constexpr (MH == 1 && NH == 2) {
                // slots A0 B0 B1; phases (0,0) (0,1); K-tile g+2 is issued while K-tile g is multiplied
                G8_READ_B(fb0, 1); G8_READ_A(0);
                G8_FENCE(); advance(); issue(0, is.kt, (g + 2) % RING_KT); issue(1, is.kt, (g + 2) % RING_KT); G8_VMCNT(10);
                G8_BARRIER(); G8_MMA(acc[0][0], fb0); G8_CS(0); G8_BARRIER();
                G8_READ_B(fb1, 2);
                G8_FENCE(); issue(2, is.kt, (g + 2) % RING_KT); G8_VMCNT(8);
                G8_BARRIER(); G8_MMA(acc[0][1], fb1); G8_BARRIER();
            } else if constexpr (MH == 2 && NH == 1) {
                // slots A0 B0 A1; phases (0,0) (1,0)
                G8_READ_B(fb0, 1); G8_READ_A(0);
                G8_FENCE(); advance(); issue(0, is.kt, (g + 2) % RING_KT); issue(1, is.kt, (g + 2) % RING_KT); G8_VMCNT(10);
                G8_BARRIER(); G8_MMA(acc[0][0], fb0); G8_CS(0); G8_BARRIER();
                G8_READ_A(2);
                G8_FENCE(); issue(3, is.kt, (g + 2) % RING_KT); G8_VMCNT(8);
                G8_BARRIER(); G8_MMA(acc[1][0], fb0); G8_CS(MH - 1); G8_BARRIER();
            } else {
                // slots A0 B0; one phase; K-tile g+2 is issued while K-tile g is multiplied (ring of 4)
                G8_READ_B(fb0, 1); G8_READ_A(0);
                G8_FENCE(); advance(); issue(0, is.kt, (g + 2) % RING_KT); issue(1, is.kt, (g + 2) % RING_KT); G8_VMCNT(4);
                G8_BARRIER(); G8_MMA(acc[0][0], fb0); G8_CS(0); G8_BARRIER();
            }
#undef G8_CS
#undef G8_READ_A
#undef G8_READ_B
#undef G8_MMA
            // the bias loads of this tile are waited for HERE, one K-tile after they were issued (an opaque use: hipcc does
            // not know about the LDS-DMA in flight and waits vmcnt(0)): by now the previous tile's stores have completed and
            // only ring traffic that the next phases need anyway is outstanding; at the epilogue the values are just there
            if constexpr ((EPI & MVLT_EPI_BIAS) != 0) {
                if (kt == un.kt0) {
#pragma unroll
                    for (int c = 0; c < NH; ++c)
#pragma unroll
                        for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(bias_v[c][j]));
                }
            }
        }
        // ---- epilogue of this unit; the next unit's first half-tiles are already on their way
        // acc[r] <-> n = nb + 4 * (lane >> 4) + r, m = mb + (lane & 15)   (MFMA issued as (B fragment, A fragment))
        if (ord == 0) G8_STAMP(4);
        bool finish = true;
        if (S > 1) {
            // k-slices of one tile meet through f32 slabs; the LAST arriver sums them in slice order and stores the tile
            // (cdna_hip_programming.md, in-launch split-K: plain stores -> every wave's vmcnt(0) -> workgroup barrier ->
            // lane 0: agent release, vmcnt(0), relaxed ticket; last arriver: agent acquire, vmcnt(0), barrier, plain loads).
            // The two wave groups run one barrier apart: group 0 takes one extra barrier here so that the barriers below
            // mean the same program point for all 8 waves; group 1 takes one behind the epilogue to fall back again.
            if (wr == 0) G8_BARRIER();
            constexpr int TILE_F = BM * BN;
            float* slab = gp.slabs + ((long)un.tile * S + un.slice) * TILE_F + wave * (TILE_F / 8);
#pragma unroll
            for (int a = 0; a < MH; ++a)
#pragma unroll
                for (int c = 0; c < NH; ++c)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            store4f(slab + ((((a * NH + c) * 4 + i) * 2 + j) * 64 + lane) * 4, acc[a][c][i][j]);
            if constexpr (AKM && BKM) {
                if (lane < 16) {
#pragma unroll
                    for (int a = 0; a < MH; ++a)
                        gp.cs_slabs[((long)un.tile * S + un.slice) * BM + a * 128 + wr * 64 + wc * 16 + lane] = cs[a][0];
                }
            }
            G8_VMCNT(0);
            G8_BARRIER();
            if (threadIdx.x == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                G8_VMCNT(0);
                eff_lds[8] = __hip_atomic_fetch_add(gp.counters + un.tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            G8_BARRIER();
            finish = __builtin_amdgcn_readfirstlane(eff_lds[8]) == S - 1;
            if (finish) {
                if (threadIdx.x == 0) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); G8_VMCNT(0); }
                G8_BARRIER();
#pragma unroll
                for (int a = 0; a < MH; ++a)
#pragma unroll
                    for (int c = 0; c < NH; ++c)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) acc[a][c][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                for (int sl = 0; sl < S; ++sl) {
                    const float* src = gp.slabs + ((long)un.tile * S + sl) * TILE_F + wave * (TILE_F / 8);
#pragma unroll
                    for (int a = 0; a < MH; ++a)
#pragma unroll
                        for (int c = 0; c < NH; ++c)
#pragma unroll
                            for (int i = 0; i < 4; ++i)
#pragma unroll
                                for (int j = 0; j < 2; ++j)
                                    acc[a][c][i][j] += load4f(src + ((((a * NH + c) * 4 + i) * 2 + j) * 64 + lane) * 4);
                }
                if constexpr (AKM && BKM) {
#pragma unroll
                    for (int a = 0; a < MH; ++a) {
                        float t = 0.f;
                        if (lane < 16)
                            for (int sl = 0; sl < S; ++sl)
                                t += gp.cs_slabs[((long)un.tile * S + sl) * BM + a * 128 + wr * 64 + wc * 16 + lane];
                        cs[a][0] = t;
                    }
                }
            }
        }
        if (finish) {
            if constexpr (WIDE) g8_tile_epilogue_wide<MH, NH, EPI>(p, by * BM + wr * 64, bx * BN + wc * 32, lane, acc, bias_v);
            else g8_tile_epilogue<MH, NH, EPI>(p, by * BM + wr * 64, bx * BN + wc * 32, lane, acc, bias_v);
            if constexpr (AKM && BKM) {
                float* db = gp.colsum[0];
#pragma unroll
                for (int i = 1; i < G8_GROUP_MAX; ++i) if (i == un.item) db = gp.colsum[i];
                if (db && bx == 0 && lane < 16) {
#pragma unroll
                    for (int a = 0; a < MH; ++a) {
                        const int m = by * BM + a * 128 + wr * 64 + wc * 16 + lane;
                        if (m < p.M) db[m] = cs[a][0];
                    }
                }
            }
        }
        if (ord == 0) G8_STAMP(5);
        if (S > 1 && wr == 1) G8_BARRIER();
    }
    if (wr == 0) G8_BARRIER();                                            // balance group 1's extra barrier
    G8_STAMP(6);
    G8_VMCNT(0);                                                          // the overrun issues land before the LDS is released
    G8_STAMP(7);
}

// 512 zero bytes for the reduction rows beyond a ragged K (k-major A operand)
__device__ __attribute__((aligned(256))) unsigned char g8_zero_page[512];

// Persistent workgroups of a weight-gradient launch: HALF the CUs.  These launches run on the side stream beside the dgrad chain,
// and one 8-wave workgroup takes a CU's whole LDS and register file: with 256 of them the chain's kernels found no CU at all while
// a group ran (stage-0 / 1 of the B = 32 step: the chain kernel beside each of the four groups took 80-150 us instead of 25-50).
// 128 workgroups with the k-slices planned for 128 units: a group takes 122-128 us instead of 83-99 stand-alone (the reduction is
// HBM-bound, fewer CUs pull harder each), the chain keeps the other half of the chip, the step gains 0.1 ms
// (11.84 -> 11.74 ms, 3 interleaved rounds; 96 / 112 / 144 / 160 workgroups: -0.05 .. -0.08; profiles/r5_g8_wgrad_grid.txt).
int g8_wgrad_cap() {
    static const int wcap = [] { const char* e = getenv("MVLT_G8_WGRAD_GRID"); return e && atoi(e) > 0 ? atoi(e) : 128; }();
    return wcap;
}
template <int MH, int NH, bool AKM, bool BKM, int EPI, bool WIDE = false>
int g8_launch1(const G8Group& gp, long units, hipStream_t s) {
    constexpr int sh = G8Cfg<MH, NH>::LDS + 64;                           // ring + the products' effective sizes + ticket
    static const bool attr = [] { return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm8_kernel<MH, NH, AKM, BKM, EPI, WIDE>),
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, sh) == hipSuccess; }();
    if (!attr) return -1;
    const int cap = (AKM && BKM) ? g8_wgrad_cap() : 256;
    const int grid = units < cap ? (int)units : cap;
    hipLaunchKernelGGL((gemm8_kernel<MH, NH, AKM, BKM, EPI, WIDE>), dim3(grid), dim3(512), sh, s, gp);
    return hipGetLastError() == hipSuccess ? 1 : -1;
}
// forward / dgrad products with bf16 rows out: the 16-byte epilogue when EVERY product of the list qualifies (GemmDev.wide)
template <int MH, int NH, bool AKM, bool BKM, int EPI>
int g8_launch(const G8Group& gp, long units, hipStream_t s) {
    if constexpr (!AKM && (EPI & MVLT_EPI_OUT_F32) == 0) {
        const char* we = getenv("MVLT_G8_WIDE"); const bool wide_on = !we || atoi(we) != 0;   // (A/B switch)
        bool wide = wide_on;
        for (int i = 0; i < gp.n; ++i) wide = wide && gp.g[i].wide;
        if (wide) return g8_launch1<MH, NH, AKM, BKM, EPI, true>(gp, units, s);
    }
    return g8_launch1<MH, NH, AKM, BKM, EPI, false>(gp, units, s);
}

const void* g8_zero_ptr() {
    static const void* ptr = [] { void* q = nullptr; return hipGetSymbolAddress(&q, HIP_SYMBOL(g8_zero_page)) == hipSuccess ? q : nullptr; }();
    return ptr;
}

template <int MH, int NH>
int g8_dispatch(const G8Group& gp, bool akm, bool bkm, int epi, long units, hipStream_t s) {
    constexpr int B_ = MVLT_EPI_BIAS, G_ = MVLT_EPI_GELU, P_ = MVLT_EPI_SAVE_PRE, X_ = MVLT_EPI_MUL_GELU_GRAD,
                  R_ = MVLT_EPI_RESIDUAL, F_ = MVLT_EPI_OUT_F32;
    if (akm && bkm && epi == F_) return g8_launch<MH, NH, true, true, F_>(gp, units, s);
    if constexpr (MH + NH > 2) {
        if (!akm && !bkm) {
            switch (epi) {
                case 0: return g8_launch<MH, NH, false, false, 0>(gp, units, s);
                case B_: return g8_launch<MH, NH, false, false, B_>(gp, units, s);
                case B_ | G_: return g8_launch<MH, NH, false, false, B_ | G_>(gp, units, s);
                case B_ | G_ | P_: return g8_launch<MH, NH, false, false, B_ | G_ | P_>(gp, units, s);
                default: return 0;
            }
        }
        if (!akm && bkm) {
            switch (epi) {
                case 0: return g8_launch<MH, NH, false, true, 0>(gp, units, s);
                case X_: return g8_launch<MH, NH, false, true, X_>(gp, units, s);
                case R_: return g8_launch<MH, NH, false, true, R_>(gp, units, s);
                default: return 0;
            }
        }
    }
    return 0;
}

// tile shape for a list of forward / dgrad products: 0 = not worth it (the 4-wave kernels of gemm.hip take it).
// big_only (the automatic mode): 256 x 256 tiles for products that fill the chip with them AND either have a long
// reduction (>= 24 K-tiles: the ~16 us of launch + first loads + store tail are amortised; 4096^3: 1.3 PFLOP/s against
// 0.78 for the 4-wave kernel) or several tiles per CU.  The B = 32 step's own mid-size products (BERT FFN: 156-204 tiles,
// 12 K-tiles) stay on the 4-wave kernels: standalone the engine is 15 % faster there, inside the step it is slower --
// one 8-wave workgroup owns a CU's whole LDS, so the weight-gradient stream cannot share the CU with it.
int g8_choose(const GemmDev* d, int n, bool big_only, long* tiles_out) {
    long t22 = 0, t12 = 0;
    int kmin = d[0].K;
    for (int i = 0; i < n; ++i) {
        t22 += (long)ceil_div(d[i].M, 256) * ceil_div(d[i].N, 256);
        t12 += (long)ceil_div(d[i].M, 128) * ceil_div(d[i].N, 256);
        kmin = d[i].K < kmin ? d[i].K : kmin;
    }
    int mode = 0;
    if (big_only) mode = (t22 >= 400 || (t22 >= 200 && kmin >= 1536)) ? 22 : 0;          // (the default path: no getenv per call)
    else if (const char* e = getenv("MVLT_G8_TILE")) mode = atoi(e);          // experiments (MVLT_G8=1): 22 / 12 force a shape
    else if (t22 >= 200) mode = 22;
    else if (t12 >= 96) mode = 12;
    *tiles_out = mode == 22 ? t22 : t12;
    return mode;
}

// Weight-gradient groups (both operands k-major, reduction over the activation rows): tile shape AND number of k-slices.
// Few output tiles, thousands of reduction rows: without k-slices a BertLayer group has 108 tiles of 256 x 256 for 256
// CUs.  Model per unit: K-tiles x time per K-tile of the shape (1.5 / 0.7 / 0.45 us for 256x256 / 128x256 / 128x128,
// measured in gpurun_out g8_check) x rounds of g8_wgrad_cap() units, plus the last arriver's slab sum.
struct G8Plan { int mode, split; long tiles, units; double est_us; size_t ws_bytes; };
G8Plan g8_plan_kk(const GemmDev* d, int n) {
    G8Plan best{0, 1, 0, 0, 1e30, 0};
    int kmax = 0;
    for (int i = 0; i < n; ++i) kmax = d[i].K > kmax ? d[i].K : kmax;
    const int nk = ceil_div(kmax, 64);
    const int forced = [] { const char* e = getenv("MVLT_G8_TILE"); return e ? atoi(e) : 0; }();
    const int forced_s = [] { const char* e = getenv("MVLT_G8_SPLIT"); return e ? atoi(e) : 0; }();
    const struct { int mode, bm, bn; double tk, red; } shapes[3] = {{22, 256, 256, 1.5, 2.6}, {12, 128, 256, 0.7, 1.3}, {11, 128, 128, 0.45, 0.65}};
    for (const auto& sh : shapes) {
        if (forced && forced != sh.mode) continue;
        long tiles = 0;
        for (int i = 0; i < n; ++i) tiles += (long)ceil_div(d[i].M, sh.bm) * ceil_div(d[i].N, sh.bn);
        for (int S = 1; S <= 32; ++S) {
            if (forced_s && forced_s != S) continue;
            if (S > 1 && nk / S < 6) break;
            const long units = tiles * S;
            const double est = (double)ceil_div(units, g8_wgrad_cap()) * ceil_div(nk, S) * sh.tk + (S > 1 ? 5.0 + sh.red * S : 0.0) + 7.0;
            if (est < best.est_us) {
                best = G8Plan{sh.mode, S, tiles, units, est, 0};
                if (S > 1) best.ws_bytes = (size_t)units * sh.bm * sh.bn * 4 + (size_t)units * sh.bm * 4 + (size_t)tiles * 4 + 1024;
            }
        }
    }
    return best;
}

}  // namespace

#ifdef G8_TRACE
extern "C" int mvlt_gemm8_trace_buffer(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_g8_trace), &buf, sizeof(buf)) == hipSuccess ? MVLT_OK : MVLT_ERR_LAUNCH;
}
#endif

// Launchers used by gemm.hip's dispatch: return 1 when the product(s) were taken, 0 when not eligible (the caller then
// uses the 4-wave kernels), -1 on a launch error.  Eligible: bf16, 16-byte aligned operand rows, no split-K requested,
// 8-byte aligned epilogue operands, N a multiple of 4, an epilogue flag set that has an instantiation, and K a multiple
// of 64 unless both operands are k-major (weight gradients: the reduction runs over activation rows, any count, also
// read from m_dev).  Weight-gradient groups: `colsum[i]` (may be null) receives the bias gradient of product i;
// `ws` / `ws_bytes`: workspace for the k-slices (mvlt_gemm8_group_workspace says how much; too little: no slices).
extern "C" __attribute__((visibility("hidden"))) size_t mvlt_gemm8_group_workspace(const void* dev_blocks, int n) {
    if (n < 1 || n > G8_GROUP_MAX) return 0;
    return g8_plan_kk(reinterpret_cast<const GemmDev*>(dev_blocks), n).ws_bytes;
}

extern "C" __attribute__((visibility("hidden"))) int mvlt_gemm8_try(const void* dev_blocks, int n, int a_kmajor, int b_kmajor,
                                                                    int big_only, float* const* colsum, void* ws, size_t ws_bytes,
                                                                    void* stream) {
    const GemmDev* d = reinterpret_cast<const GemmDev*>(dev_blocks);
    if (n < 1 || n > G8_GROUP_MAX) return 0;
    G8Group gp{};
    gp.n = n;
    for (int i = 0; i < n; ++i) {
        if (!d[i].a_vec || !d[i].b_vec || d[i].split_k > 1 || !d[i].epi_vec || d[i].epi != d[0].epi || d[i].a_colsum) return 0;
        if (!(a_kmajor && b_kmajor) && (d[i].K % 64 != 0)) return 0;
        if (d[i].K < 64 || d[i].N % 4 != 0 || d[i].N < 4) return 0;
        if (a_kmajor && (d[i].M % 8 != 0)) return 0;
        if (b_kmajor && (d[i].N % 8 != 0)) return 0;
        gp.g[i] = d[i];
        gp.colsum[i] = colsum ? colsum[i] : nullptr;
    }
    gp.zero_page = g8_zero_ptr();
    if (a_kmajor && !gp.zero_page) return 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (a_kmajor && b_kmajor) {
        G8Plan pl = g8_plan_kk(d, n);
        if (pl.split > 1 && (!ws || ws_bytes < pl.ws_bytes)) {          // no room for the slabs: unsliced plan of the same shape
            pl.split = 1; pl.units = pl.tiles;
        }
        gp.split = pl.split;
        if (pl.split > 1) {
            const int bm = pl.mode == 22 ? 256 : 128, bn = pl.mode == 11 ? 128 : 256;
            char* w = reinterpret_cast<char*>(ws);
            gp.slabs = reinterpret_cast<float*>(w);
            gp.cs_slabs = reinterpret_cast<float*>(w + (size_t)pl.units * bm * bn * 4);
            gp.counters = reinterpret_cast<int*>(w + (size_t)pl.units * bm * bn * 4 + (size_t)pl.units * bm * 4);
            if (hipMemsetAsync(gp.counters, 0, (size_t)pl.tiles * 4, s) != hipSuccess) return -1;          // tickets: re-initialised every call
        }
        if (pl.mode == 22) return g8_dispatch<2, 2>(gp, true, true, d[0].epi, pl.units, s);
        if (pl.mode == 12) return g8_dispatch<1, 2>(gp, true, true, d[0].epi, pl.units, s);
        if (pl.mode == 11) return g8_dispatch<1, 1>(gp, true, true, d[0].epi, pl.units, s);
        return 0;
    }
    long tiles = 0;
    const int mode = g8_choose(d, n, big_only != 0, &tiles);
    if (mode == 22) return g8_dispatch<2, 2>(gp, a_kmajor != 0, b_kmajor != 0, d[0].epi, tiles, s);
    if (mode == 12) return g8_dispatch<1, 2>(gp, a_kmajor != 0, b_kmajor != 0, d[0].epi, tiles, s);
    return 0;
}
