// Row-streaming products for the HBM-bound nn.Linear calls of the Swin stages 0 / 1 (round 5): Mlp fc1 / fc2
// (visual_feature_extractor.py:135-141) and the dgrads of fc1 / fc2 / proj / qkv (:231, 252) at C = 96 / 192 with
// 100 k / 25 k token rows.  These products move 1.2 - 1.7 KB per row for 150 - 300 kFLOP: their floor is HBM time (fc1 of
// stage 0 at B = 32: 173 MB = 31 us at 5.5 TB/s), and the tile kernels of gemm.hip reach ~3 TB/s on them (57 us): a workgroup
// there loads a tile, multiplies for a microsecond, stores 64 KB and exits -- nothing overlaps inside it.
//
// Here the WEIGHT is stationary and the ROWS stream:
//   * one persistent workgroup per CU owns a contiguous range of token rows and walks it in STAGES of 32 rows;
//   * the whole weight matrix (<= 73.7 KB; stage 1: a 192-column slice of it, the four slices on workgroups of one XCD)
//     lives in the REGISTERS of the consumer waves as MFMA operand fragments: consumer wave (wm, wn) owns rows 16 wm .. +16
//     of every stage and FN 16-column blocks, i.e. FN x K/32 fragments (18 - 24, 72 - 96 VGPRs) loaded once;
//   * two LOADER waves do nothing but LDS-DMA (global_load_lds_dwordx4, 1 KB per instruction): the stage's activation rows
//     -- a contiguous block, lda = K -- and, for epilogues that read a row operand (residual, gelu' operand), that operand's
//     [32, N] block go into a RING of LDS slots that fills the CU's 160 KB (5 - 21 slots, up to 60 instructions = 60 KB in
//     flight per loader wave: the vmcnt field has 6 bits).  A loader's vmcnt is exact (it issues nothing else), so the only
//     waits in the kernel are one counted s_waitcnt per stage in the loaders and ONE workgroup barrier per stage;
//   * consumers never issue a global load inside the loop (their vmcnt holds stores only, which they never wait for): A
//     fragments and epilogue operands come from LDS, bias / weights sit in registers, the DropPath scale is a scalar load;
//   * outputs leave in the 8-column chunk layout (v_permlane16_swap, gemm_dev.h): 16 bytes per lane per store.
// Per stage a consumer issues K/32 ds_read_b128 and FN K/32 MFMAs -- a few hundred cycles against the 1.7 - 2.6 us the
// stage's bytes take at the HBM rate: arithmetic and LDS are off the critical path by construction.
//
// Barrier protocol (stage i lives in slot i % RING, RING = D + 1):
//   loader:   wait until its share of stage i has landed (counted vmcnt) | BARRIER i | issue stage i + D into the slot of
//             stage i - 1 (every consumer finished reading it before it arrived at barrier i)
//   consumer: BARRIER i | read stage i, multiply, epilogue
// k-major weights (dgrads: dx = dy W reads W [N_out, K_in] along its rows): the loaders first DMA the weight into the tail
// of the ring, the consumers pull their fragments out with ds_read_b64_tr_b16 (transposing read), and only then do the slots
// under the weight image join the ring (two extra barriers in the prologue).
#include "common.h"
#include "gemm_dev.h"
#include <cstdlib>

namespace {

constexpr int RS_TR = 32;                 // rows per stage
constexpr int RS_LDS_MAX = 160 * 1024;
// Default routing (bits as in mvlt_rowstream_try): the six stage-0 shapes.  Measured at B = 32 (profiles/r5_rowstream.md): stage 0
// 48.5 / 29.6 / 52.3 / 24.7 / 11.7 / 16.0 us against 52.5 / 28.5 / 50.1 / 24.4 / 16.5 / 23.1 us for the tile kernels, step 12.05 ->
// 11.97 ms (interleaved, same box); the stage-1 shapes (bits 2, 7: four column chunks per row group re-read the rows; bit 8) are
// slower or equal here and stay on the tile kernels.
constexpr unsigned RS_DEFAULT_MASK = 0x07Bu;

template <int K_, int N_, int WN_, bool BKM_, bool X2_> struct RsCfg {
    static constexpr int K = K_, N = N_, WN = WN_;
    static constexpr bool BKM = BKM_, X2 = X2_;
    static constexpr int FN = N / (16 * WN), NP = FN / 2, KS = K / 32;
    static constexpr int NCW = 2 * WN, NLW = 2, NT = (NCW + NLW) * 64;
    static constexpr int XB = RS_TR * K * 2, EB = X2 ? RS_TR * N * 2 : 0, SL = XB + EB;
    static constexpr int NIX = XB / 1024, NIE = EB / 1024, NI = NIX + NIE, NIW = NI / 2;
    static constexpr int RING_L = RS_LDS_MAX / SL, RING_V = 60 / NIW + 1;
    static constexpr int RING = RING_L < RING_V ? RING_L : RING_V, D = RING - 1;
    static constexpr int WB = K * N * 2, NIWT = WB / 1024, NIWW = NIWT / 2;
    static constexpr int SW = BKM ? (WB + SL - 1) / SL : 0;
    static constexpr int D0 = BKM ? (D < RING - SW ? D : RING - SW) : D;
    static constexpr int LDS = RING * SL;
    static_assert(K % 32 == 0 && N % (32 * WN) == 0, "fragment pairs");
    static_assert(XB % 1024 == 0 && EB % 1024 == 0 && NI % 2 == 0, "whole LDS-DMA instructions, shared by two loader waves");
    static_assert(D >= 2 && D0 >= 1 && (D - 1) * NIW <= 63 && D0 * NIW <= 63, "ring depth / vmcnt range");
    static_assert(!BKM || (WB % 2048 == 0 && NIWW <= 63 && SW * SL >= WB && RING - SW >= 1), "weight image in the ring's tail");
};

MVLT_DEV void rs_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
template <int N> MVLT_DEV void rs_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
// wait until at most rem * NIW of this wave's LDS-DMA instructions are outstanding (rem <= MAXR, wave-uniform)
template <int NIW, int MAXR> MVLT_DEV void rs_wait_rem(int rem) {
    if constexpr (MAXR == 0) rs_vmcnt<0>();
    else { if (rem >= MAXR) rs_vmcnt<(MAXR * NIW)>(); else rs_wait_rem<NIW, MAXR - 1>(rem); }
}

// ---- bank-conflict-free row images (round 6).  A stage's rows land in LDS by LDS-DMA, which writes base + 16 lane: the image is
// linear, [32 rows][CPR 16-byte chunks].  A consumer's ds_read_b128 of an MFMA fragment takes chunk 4 kb + g of rows mr = 0..15;
// with a row stride of K * 2 bytes = 768 (K = 384) all 16 rows start on the same bank: SQ_LDS_BANK_CONFLICT was 87 % of
// SQ_LDS_IDX_ACTIVE on the K = 384 shapes and exactly 50 % on K = 96 / 288 (profiles/r5_step_rowstream_pmc.txt).  As in
// gemm8.hip the fix sits on the DMA's SOURCE side: position `pos` of row r receives logical chunk pos ^ s(r) (inside aligned
// groups of GRP = 16 / 8 / 4 chunks, whatever divides the row), and the reader asks for position c ^ s(r).  s() per row
// width and per reader (the A fragment reads chunks in lane-group order g, the epilogue operand in the chunk-layout order
// 2 (g & 1) + (g >> 1)) was chosen with scripts/lds_bank_model.py: every read below costs the ideal 4 LDS cycles.
template <int CPR> struct RsSwz {
    static constexpr int GRP = CPR % 16 == 0 ? 16 : (CPR % 8 == 0 ? 8 : 4);
    static constexpr int NQ = GRP / 4;
    template <bool EPI> static MVLT_DEV int s(int r) {
        if constexpr (GRP == 16) return r & 15;
        else if constexpr (GRP == 8) return EPI ? (2 * ((r >> 1) & 3) + ((r >> 3) & 1)) : ((r >> 1) & 7);
        else return ((EPI ? 0xB4 : 0x78) >> (2 * ((r >> 2) & 3))) & 3;          // EPI: [0, 1, 3, 2], A: [0, 2, 3, 1] by (r >> 2) & 3
    }
    // logical chunk that position `pos` of row r holds (= position of logical chunk `pos`: the map is an involution)
    template <bool EPI> static MVLT_DEV int chunk(int r, int pos) { return (pos & ~(GRP - 1)) | ((pos ^ s<EPI>(r)) & (GRP - 1)); }
};

// k-major weight image [K][N] (dgrads), read once per workgroup by the transposing ds_read_b64_tr_b16: the 16 k-rows of a fragment
// half sit N * 2 bytes apart (768 B at N = 384: all on one bank group, 8-way conflicts).  Row k's 16-byte chunks are XOR-ed
// with wswz(k) (even: the two chunks a lane group reads stay neighbours); same model, every read at the ideal 2 cycles.
template <int N> MVLT_DEV int rs_wswz(int k) {
    constexpr int CPR = N / 8;
    if constexpr (CPR % 16 == 0) return 2 * (k & 3) + 8 * ((k >> 3) & 1);
    else if constexpr (CPR % 8 == 0) return 2 * ((k >> 1) & 1) + 4 * ((k >> 3) & 1);
    else return 2 * ((k >> 3) & 1);
}
template <int N> MVLT_DEV bf16x8 rs_frag_kmajor(const bf16_t* wl, int row0, int k0) {
    const int l = threadIdx.x & 63;
    const int g = l >> 4, i = l & 15, q = i >> 2, pp = i & 3;
    const int ka = k0 + 8 * g + q, kb = ka + 4, n = row0 + 4 * pp, ch = n >> 3, w = n & 7;
    const bf16_t* p0 = wl + ka * N + ((ch ^ rs_wswz<N>(ka)) << 3) + w;
    const bf16_t* p1 = wl + kb * N + ((ch ^ rs_wswz<N>(kb)) << 3) + w;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p0);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p1);
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

struct RsArgs { GemmDev g; int n_chunks; };

template <int K, int N, int WN, bool BKM, bool X2>
__global__ __launch_bounds__((2 * WN + 2) * 64) void rowstream_kernel(const RsArgs a) {
    using Cfg = RsCfg<K, N, WN, BKM, X2>;
    constexpr int FN = Cfg::FN, NP = Cfg::NP, KS = Cfg::KS, NCW = Cfg::NCW, SL = Cfg::SL, XB = Cfg::XB;
    constexpr int RING = Cfg::RING, D = Cfg::D, D0 = Cfg::D0, NIW = Cfg::NIW, NIX = Cfg::NIX;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const GemmDev& p = a.g;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int rows = p.M;

    // ---- this workgroup's (column chunk, row range).  Blocks b, b + 8, b + 16, ... share an XCD (observed round-robin
    // placement; speed only): the chunks of one row group are such neighbours, so the rows they all read meet in one L2.
    const int nch = a.n_chunks, G = gridDim.x, b = blockIdx.x;
    const int chunk = nch > 1 ? (b >> 3) % nch : 0;
    const int rg = nch > 1 ? (b & 7) + 8 * (b / (8 * nch)) : b;
    const int RG = G / nch;
    const int F = (rows + 15) >> 4;                                   // 16-row fragments
    const int f0 = (int)((long)F * rg / RG), f1 = (int)((long)F * (rg + 1) / RG);
    const int r_begin = f0 * 16, r_end = min(f1 * 16, rows);
    const int ns = (r_end - r_begin + RS_TR - 1) / RS_TR;             // stages of this workgroup (wave-uniform)
    const int col0 = chunk * N;                                       // first output column of the chunk
    const bf16_t* X = reinterpret_cast<const bf16_t*>(p.A);
    const bf16_t* W = reinterpret_cast<const bf16_t*>(p.B);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    if (wave >= NCW) {
        // =========================================================================================== loader waves
        const int lw = wave - NCW;
        const bf16_t* E = nullptr; long lde = 0;
        if constexpr (X2) {
            if (p.epi & MVLT_EPI_MUL_GELU_GRAD) { E = reinterpret_cast<const bf16_t*>(p.aux); lde = p.ldc; }
            else { E = reinterpret_cast<const bf16_t*>(p.residual); lde = p.ldr; }
        }
        const long xlast = (long)rows * K - 8;                        // last whole 16-byte chunk of X
        auto issue = [&](int s) {
            const int r0 = r_begin + s * RS_TR;
            const unsigned slot = lds0 + (unsigned)(s % RING) * SL;
#pragma unroll
            for (int j = 0; j < NIW; ++j) {
                const int idx = 2 * j + lw;                           // this wave's instruction of the stage
                if (idx < NIX) {                                      // activation rows: one contiguous block, chunks swizzled inside each row
                    const int q = idx * 64 + lane, r = q / (K / 8), pos = q - r * (K / 8);
                    const long off = (long)(r0 + r) * K + RsSwz<K / 8>::template chunk<false>(r, pos) * 8;
                    glds16_asm(X + (off < xlast ? off : xlast), slot + idx * 1024);
                } else if constexpr (X2) {                            // row operand of the epilogue: [32, N] of an ld-strided matrix
                    const int q = (idx - NIX) * 64 + lane, r = q / (N / 8), pos = q - r * (N / 8);
                    glds16_asm(E + (long)min(r0 + r, rows - 1) * lde + col0 + RsSwz<N / 8>::template chunk<true>(r, pos) * 8, slot + idx * 1024);
                }
            }
        };
        if constexpr (BKM) {
            // weight image [K][N] (row = one k, N output columns of this chunk) into the ring's tail
            constexpr int NIWW = Cfg::NIWW;
            const unsigned wdst = lds0 + (unsigned)(RING - Cfg::SW) * SL;
#pragma unroll
            for (int j = 0; j < NIWW; ++j) {
                const int idx = 2 * j + lw, q = idx * 64 + lane, k = q / (N / 8), pos = q - k * (N / 8);
                glds16_asm(W + (long)k * p.ldb + col0 + (pos ^ rs_wswz<N>(k)) * 8, wdst + idx * 1024);
            }
            for (int s = 0; s < D0; ++s) if (s < ns) issue(s);
            rs_wait_rem<NIW, D0>(ns < D0 ? ns : D0);                  // the weight image has landed
            rs_barrier();                                             // P1: consumers read their fragments out
            rs_barrier();                                             // P2: the image's slots join the ring
            for (int s = D0; s < D; ++s) if (s < ns) issue(s);
        } else {
            for (int s = 0; s < D; ++s) if (s < ns) issue(s);
        }
        for (int i = 0; i < ns; ++i) {
            const int rem = ns - 1 - i;
            rs_wait_rem<NIW, D - 1>(rem < D - 1 ? rem : D - 1);       // this wave's share of stage i has landed
            rs_barrier();
            if (i + D < ns) issue(i + D);
        }
        return;
    }

    // =============================================================================================== consumer waves
    const int wm = wave / WN, wn = wave - wm * WN;
    const int mr = lane & 15, g = lane >> 4;
    const int cofs = 16 * (g & 1) + 8 * (g >> 1);                     // chunk layout: this lane's 8 columns inside a fragment pair
    const int ncol = wn * 16 * FN;                                    // first column of this wave inside the chunk
    bf16x8 fw[FN][KS];
    if constexpr (BKM) {
        rs_barrier();                                                 // P1
        const bf16_t* wl = reinterpret_cast<const bf16_t*>(smem + (RING - Cfg::SW) * SL);
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int kb = 0; kb < KS; ++kb) fw[j][kb] = rs_frag_kmajor<N>(wl, ncol + 16 * j, 32 * kb);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        rs_barrier();                                                 // P2
    } else {
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int kb = 0; kb < KS; ++kb)
                fw[j][kb] = *reinterpret_cast<const bf16x8*>(W + (long)(col0 + ncol + 16 * j + mr) * p.ldb + 32 * kb + 8 * g);
    }
    // byte offset of chunk 4 kb + g of this lane's row inside its aligned chunk group, for the NQ values of kb mod NQ
    using SA = RsSwz<K / 8>;
    using SE = RsSwz<N / 8>;
    int xq[SA::NQ];
    {
        const int sa = SA::template s<false>(mr);
#pragma unroll
        for (int i = 0; i < SA::NQ; ++i) xq[i] = ((i ^ (sa >> 2)) & (SA::NQ - 1)) * 64 + ((g ^ sa) & 3) * 16;
    }
    const int se = SE::template s<true>(mr);
    const int epi = p.epi;
    f32x4 bias_lo[NP], bias_hi[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        bias_lo[q] = f32x4{0.f, 0.f, 0.f, 0.f}; bias_hi[q] = bias_lo[q];
        if (epi & MVLT_EPI_BIAS) {
            const float* bp = p.bias + col0 + ncol + 32 * q + cofs;
            bias_lo[q] = *reinterpret_cast<const f32x4*>(bp); bias_hi[q] = *reinterpret_cast<const f32x4*>(bp + 4);
        }
    }
    bf16_t* Y = reinterpret_cast<bf16_t*>(p.C);
    bf16_t* PRE = reinterpret_cast<bf16_t*>(p.pre);
    const bool do_gelu = (epi & MVLT_EPI_GELU) != 0, do_pre = do_gelu && (epi & MVLT_EPI_SAVE_PRE) != 0;
    const bool do_aux = (epi & MVLT_EPI_MUL_GELU_GRAD) != 0, do_res = (epi & MVLT_EPI_RESIDUAL) != 0, do_scale = (epi & MVLT_EPI_ROWSCALE) != 0;

    for (int i = 0; i < ns; ++i) {
        rs_barrier();
        const char* slot = smem + (i % RING) * SL;
        const int r0 = r_begin + i * RS_TR + 16 * wm;                 // first row of this wave's fragment (wave-uniform)
        f32x4 acc[FN];
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* xa = slot + (16 * wm + mr) * K * 2;
#pragma unroll
        for (int kb = 0; kb < KS; ++kb) {
            const bf16x8 fx = *reinterpret_cast<const bf16x8*>(xa + (kb & ~(SA::NQ - 1)) * 64 + xq[kb & (SA::NQ - 1)]);
#ifndef RS_ABL_NOMMA
#pragma unroll
            for (int j = 0; j < FN; ++j) Mma<bf16_t>::mma(acc[j], fw[j][kb], fx);
#else
            asm volatile("" :: "v"(fx));
#endif
        }
        float sc = 1.0f;
        if (do_scale) {
            // DropPath scale of the image this fragment's 16 rows belong to (rps is a multiple of 16): a SCALAR load -- a vector
            // load would sit in vmcnt behind this wave's stores of the previous stages (in-order) and wait for them
            const float* sp = p.rowscale + __builtin_amdgcn_readfirstlane(min(r0, rows - 1) / p.rps);
            asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(sc) : "s"(sp) : "memory");
        }
        const int m = r0 + mr;
        const bool live = m < r_end;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            f32x4 lo = acc[2 * q], hi = acc[2 * q + 1];
            swap16(lo, hi);
            lo += bias_lo[q]; hi += bias_hi[q];
            const long co = (long)m * p.ldc + col0 + ncol + 32 * q + cofs;
            if (do_gelu) {
#ifndef RS_ABL_NOSTORE
                if (do_pre && live) *reinterpret_cast<u32x4*>(PRE + co) = pack8(lo, hi);
#endif
#ifndef RS_ABL_NOGELU          /* timing-only ablation builds (scripts/rowstream_ablate.sh): outputs wrong */
#pragma unroll
                for (int e = 0; e < 4; ++e) { lo[e] = gelu_f(lo[e]); hi[e] = gelu_f(hi[e]); }
#endif
            }
            if (do_scale) { lo *= sc; hi *= sc; }
            if constexpr (X2) {
                // logical chunk 4 Q + cc of the row (Q = this fragment pair's 32-column quad, cc = cofs / 8) sits at position chunk ^ se
                const int Q = (ncol >> 5) + q, cc = cofs >> 3;
                const int epos = ((Q & ~(SE::NQ - 1)) | ((Q ^ (se >> 2)) & (SE::NQ - 1))) * 4 + ((cc ^ se) & 3);
                const u32x4 ev = *reinterpret_cast<const u32x4*>(slot + XB + (16 * wm + mr) * N * 2 + epos * 16);
                f32x4 elo, ehi;
                unpack8(ev, elo, ehi);
                if (do_aux) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { lo[e] *= gelu_grad_f(elo[e]); hi[e] *= gelu_grad_f(ehi[e]); }
                } else if (do_res) { lo += elo; hi += ehi; }
            }
#ifndef RS_ABL_NOSTORE
            if (live) *reinterpret_cast<u32x4*>(Y + co) = pack8(lo, hi);
#else
            { u32x4 keep = pack8(lo, hi); asm volatile("" :: "v"(keep)); }
#endif
        }
    }
}

template <int K, int N, int WN, bool BKM, bool X2>
int rs_launch(const RsArgs& a, hipStream_t s) {
    using Cfg = RsCfg<K, N, WN, BKM, X2>;
    auto k = rowstream_kernel<K, N, WN, BKM, X2>;
    static const bool attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS) == hipSuccess;
    if (!attr) return -1;
    static const int ncu = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
    const int unit = 8 * a.n_chunks;
    const int grid = ncu / unit * unit;
    if (grid < unit) return 0;
    hipLaunchKernelGGL(k, dim3(grid), dim3(Cfg::NT), Cfg::LDS, s, a);
    return hipGetLastError() == hipSuccess ? 1 : -1;
}

}  // namespace

// Launcher used by gemm.hip's dispatch: 1 = taken, 0 = not eligible (the tile kernels run), -1 = launch error.
// Eligible: bf16, A [M, K] contiguous rows (lda == K), one of the Swin stage-0 / 1 shapes below, enough rows (fewer leave the
// 256 persistent workgroups without work: the tile kernels win), 16-byte aligned operands, an epilogue made of
// bias / GELU (+ saved pre-activation) / DropPath row scale / gelu' operand / residual only.
extern "C" __attribute__((visibility("hidden"))) int mvlt_rowstream_try(const void* dev_block, int b_kmajor, void* stream) {
    const GemmDev& d = *reinterpret_cast<const GemmDev*>(dev_block);
    constexpr int ALLOWED = MVLT_EPI_BIAS | MVLT_EPI_GELU | MVLT_EPI_SAVE_PRE | MVLT_EPI_ROWSCALE | MVLT_EPI_MUL_GELU_GRAD | MVLT_EPI_RESIDUAL;
    if ((d.epi & ~ALLOWED) || d.m_dev || d.split_k > 1 || d.a_colsum || d.a_kmajor) return 0;
    if (d.lda != d.K || d.ldc != d.N || !d.a_vec || !d.b_vec || !d.epi_vec || d.M < 16384) return 0;
    if ((d.epi & MVLT_EPI_BIAS) && !aligned16(d.bias)) return 0;
    if ((d.epi & MVLT_EPI_MUL_GELU_GRAD) && (d.epi & MVLT_EPI_RESIDUAL)) return 0;
    if ((d.epi & MVLT_EPI_ROWSCALE) && (d.rps < 16 || d.rps % 16)) return 0;
    if ((d.epi & MVLT_EPI_RESIDUAL) && (d.ldr % 8 || !aligned16(d.residual))) return 0;
    if (b_kmajor ? d.ldb != d.N : d.ldb != d.K) return 0;
    const bool x2 = (d.epi & (MVLT_EPI_MUL_GELU_GRAD | MVLT_EPI_RESIDUAL)) != 0;
    RsArgs a{d, 1};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int K = d.K, N = d.N;
    if (!(K == 96 || K == 192 || K == 288 || K == 384) || !(N == 96 || N == 192 || N == 384 || N == 768)) return 0;
    // MVLT_ROWSTREAM = bit mask of the shapes that take this kernel (bit order below; 0 = never).  Read per call, but only
    // by the dozen products per step whose shape got this far (the tests switch shapes on and off inside one process).
    const char* env = getenv("MVLT_ROWSTREAM");
    const unsigned mask = env ? (unsigned)strtoul(env, nullptr, 0) : RS_DEFAULT_MASK;
    // default routing only from 49,152 rows (stage 0 at B >= 16: at least four 32-row stages per workgroup behind the
    // weight-to-register prologue; measured at 100,352 rows); an explicit mask (tests, experiments) routes from 16,384 rows
    if (!env && d.M < 49152) return 0;
    auto on = [&](int bit) { return (mask >> bit) & 1u; };
    if (!b_kmajor) {
        if (K == 96 && N == 384 && !x2 && on(0)) return rs_launch<96, 384, 4, false, false>(a, s);
        if (K == 384 && N == 96 && x2 && on(1)) return rs_launch<384, 96, 3, false, true>(a, s);
        if (K == 192 && N == 768 && !x2 && on(2)) { a.n_chunks = 4; return rs_launch<192, 192, 3, false, false>(a, s); }
    } else {
        if (K == 96 && N == 384 && x2 && on(3)) return rs_launch<96, 384, 4, true, true>(a, s);
        if (K == 384 && N == 96 && !x2 && on(4)) return rs_launch<384, 96, 3, true, false>(a, s);
        if (K == 96 && N == 96 && !x2 && on(5)) return rs_launch<96, 96, 3, true, false>(a, s);
        if (K == 288 && N == 96 && !x2 && on(6)) return rs_launch<288, 96, 3, true, false>(a, s);
        if (K == 192 && N == 768 && x2 && on(7)) { a.n_chunks = 4; return rs_launch<192, 192, 3, true, true>(a, s); }
        if (K == 192 && N == 192 && !x2 && on(8)) return rs_launch<192, 192, 3, true, false>(a, s);
    }
    return 0;
}
