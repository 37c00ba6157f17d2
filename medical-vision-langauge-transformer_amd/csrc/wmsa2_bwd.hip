// Backward of the attention half of a Swin block in ONE launch, in the shape of the second forward design (round 6):
// output-projection dgrad + attention backward + qkv dgrad (visual_feature_extractor.py:224-254 backward), bf16 storage /
// f32 accumulation.  Everything is in WINDOW order (dy_win, qkv_win, dqkv, dxn): no row maps in here, the LayerNorm
// backward that follows applies them.
//
// A UNIT is (two consecutive windows = 98 rows = seven 16-row MFMA tiles) x (a group of 3 heads), as in wmsa2.hip: the
// nH / 3 head groups of a window pair are different workgroups (stage 2 at B = 32: 64 pairs x 4 groups = 256 workgroups, one
// per CU).  Unlike the forward pass the groups never meet inside the launch: the qkv dgrad dXn = dQKV Wqkv sums over the
// heads, so every group writes the PARTIAL product of its own 288 columns of dQKV, dxn[hg] = dQKV[:, group] Wqkv[group, :]
// (bf16, [nH / 3][rows, C]), and the consumer -- LayerNorm backward, MvltLayerNormBwd.dy_parts -- adds the partial rows in
// f32 while it loads them.  No hand-off, no flags, no co-residency requirement.
//
// Phases of a unit (8 waves):
//   1  dO_g [112, 96] = dY [112, C] Wproj[:, 96 columns of the group]: the 98 rows of dY (one contiguous block) go through
//      registers into an XOR-swizzled LDS tile (conflict-free b128 A fragments), the weight rows through a 4-deep LDS-DMA
//      ring of single k-steps (k-major images read with ds_read_b64_tr_b16, 32-byte units swizzled on the SOURCE side);
//   2  attention backward of the 6 (window, head) problems: the two 4-wave halves of the workgroup take one window each
//      and walk the three heads with the scores-once code of attn.hip (phase A: dQ, dBias, P and dS images; phase B: dK,
//      dV through the transposing reads); Q / K / V / lse of the next problem are in flight during the current one;
//      dq / dk / dv go to HBM (the qkv weight gradient reads them) AND into an LDS tile [112, 288];
//   3  partial dXn [112, C] = that tile x Wqkv[288 rows of the group, :] with the weight rows through a 3-deep LDS-DMA ring;
//      bf16 rows -> dxn[hg].
// Relative-position-bias gradients: per-lane sums over all units of the workgroup, flushed once (atomics on the f32 table).
#include "common.h"
#include <hip/hip_ext.h>
#include "attn_frag.h"
#include <stdlib.h>

namespace {
using namespace mvlt_attn;
typedef bf16_t T;
using M = Mma<T>;

struct Wb2Dev {
    int nwin, nW, res, shift, nH, nunits;
    const T* dy; const T* qkv; const float* lse;
    const T* wproj; const T* wqkv;
    const float* bias_table; float scale;
    T* dqkv; T* dxn; long part_stride;          // elements between the partial dXn tensors
    float* dbias_ws;                            // [grid][G * 169] per-workgroup sums of dS along the relative-position diagonals
};

constexpr int B2_LD = 40, B2_LDP = 72;                           // row strides of the [64][32] and [64][64] images (as attn.hip)
constexpr int B2_IMG = 64 * B2_LD * 2, B2_PIMG = 64 * B2_LDP * 2;
constexpr int B2_HALF = 3 * B2_IMG + 2 * B2_PIMG + 256;          // Q, K, dO, P, dS images + lse of one 4-wave half
constexpr float LOG2E_ = 1.4426950408889634f;

template <int C, int G> struct Wb2Geom {
    static constexpr int M = 98, MT = 7, MR = 112;
    static constexpr int NH = C / 32, NHG = NH / G, OC = 32 * G, QC = 3 * OC;
    // k-major weight images: rows padded to a multiple of 256 bytes (8 units of 32 bytes: the swizzle's period)
    static constexpr int KB1 = C % 64 == 0 ? 64 : 32;            // k-rows per phase-1 ring stage
    static constexpr int LD1 = (OC + 127) / 128 * 128, ST1 = KB1 * LD1 * 2, NST1 = 4;     // phase 1: [KB1 k][OC], elements / bytes
    static constexpr int LD3 = (C + 127) / 128 * 128, ST3 = 32 * LD3 * 2, NST3 = 3;       // phase 3: [32 k][C]
    static constexpr int YB = MR * C * 2;                        // dY tile
    static constexpr int OB = MR * OC * 2;                       // dO tile (lives behind the images of phase 2)
    static constexpr int P2 = 2 * B2_HALF + OB;
    static constexpr int A0 = YB > P2 ? YB : P2;
    // region A: dY tile | images + dO tile | phase-3 ring (slots 1, 2 at the front, slot 0 at the END, behind the images: it is
    // filled while the last attention problem runs, when only the dO tile -- dead by then -- lives there) | partial-dXn tile
    static constexpr int A1 = 2 * B2_HALF + ST3 > 2 * ST3 ? 2 * B2_HALF + ST3 : 2 * ST3 + ST3;
    static constexpr int AB = A0 > A1 ? A0 : A1;
    static constexpr int SLOT0 = AB - ST3;
    static_assert(SLOT0 >= 2 * B2_HALF && SLOT0 >= 2 * ST3, "phase-3 ring slot 0 behind the images and the other slots");
    static constexpr int QB0 = MR * QC * 2;                      // dqkv tile
    static constexpr int QB = QB0 > NST1 * ST1 ? QB0 : NST1 * ST1;          // region B: phase-1 ring | dqkv tile
    static constexpr int TB = G * 176 * 4;
    static constexpr int bytes = AB + QB + TB;
    static_assert(bytes <= 160 * 1024, "LDS");
    static_assert(NH % G == 0 && C % 32 == 0, "head groups");
};

// [rows][CC] bf16 tile, 16-byte chunks XOR-swizzled with the row (wmsa2.hip)
template <int CC> MVLT_DEV int xoff(int row, int chunk) {
    constexpr int CPR = CC / 8;
    constexpr int GRP = CPR % 16 == 0 ? 16 : (CPR % 8 == 0 ? 8 : 4);
    return row * (CC * 2) + (((chunk & ~(GRP - 1)) | ((chunk ^ row) & (GRP - 1))) << 4);
}
MVLT_DEV int kswz8(int k) { return (k & 3) | ((k >> 1) & 4); }
typedef __attribute__((address_space(3))) char lds_char;
MVLT_DEV uint32_t lds_addr(const void* q) { return (uint32_t)(uintptr_t)(lds_char*)q; }
MVLT_DEV void glds16(const void* gsrc, unsigned lds_dst) {          // LDS-DMA, 16 bytes per lane to lds_dst + 16 lane
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> MVLT_DEV void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// first-operand fragment (rows n0 .. n0 + 15, k-slots 8 g .. 8 g + 7 of the 32 k-rows) of a k-major image [32][LD] whose 32-byte
// units are swizzled by kswz8(k) -- the layout the ring fills below produce
template <int LD> MVLT_DEV bf16x8 kfrag(const char* img, int n0) {
    const int l = threadIdx.x & 63;
    const int g = l >> 4, i = l & 15, q = i >> 2, pp = i & 3;
    const int k = 8 * g + q, c = n0 >> 4;
    const char* p0 = img + k * (LD * 2) + ((c ^ kswz8(k)) << 5) + 8 * pp;
    const char* p1 = img + (k + 4) * (LD * 2) + ((c ^ kswz8(k + 4)) << 5) + 8 * pp;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p0);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p1);
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
// one 1-KB piece (64 lanes x 16 bytes) of a k-major ring stage: image row r = 32-row block row, `cols` valid elements per row at
// src + r * src_ld; pad chunks (beyond cols) re-read chunk 0 (never used by a fragment read)
template <int LD> MVLT_DEV void ring_piece(const T* src, long src_ld, int cols, int piece, int lane, unsigned dst_stage) {
    constexpr int CPRW = LD / 8;                            // 16-byte chunks per image row
    const int f = 64 * piece + lane, r = f / CPRW, x = f - r * CPRW;
    const int ch = (((x >> 1) ^ kswz8(r)) << 1) | (x & 1);
    glds16(src + (long)r * src_ld + (ch * 8 < cols ? ch * 8 : 0), dst_stage + piece * 1024);
}

// -DWB2_TRACE (diagnostic build only): thread 0 of every workgroup stamps the 100 MHz real-time counter at the phase boundaries of
// its first unit into a buffer set by mvlt_swin_wmsa2_bwd_trace_buffer (scripts/wmsa2_bwd_trace.py reads it)
#ifdef WB2_TRACE
__device__ long long* g_wb2_trace = nullptr;
#define WB2_STAMP(k) do { if (threadIdx.x == 0 && g_wb2_trace && unit < (int)gridDim.x) g_wb2_trace[blockIdx.x * 32 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define WB2_STAMP(k) do { } while (0)
#endif

template <int C, int G>
__global__ __launch_bounds__(512) void wmsa2_bwd_kernel(const Wb2Dev p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using GM = Wb2Geom<C, G>;
    constexpr int MR = GM::MR, OC = GM::OC, QC = GM::QC, NHG = GM::NHG;
    constexpr int KS1 = C / 32, KS3 = QC / 32;
    constexpr int NT3 = C / 16, CT3 = (NT3 + 7) / 8;           // phase-3 column tiles, per wave
    constexpr int PC1 = GM::ST1 / 1024, PC3 = GM::ST3 / 1024;  // 1-KB pieces per ring stage
    constexpr int PW1 = PC1 / 8, PW3 = PC3 / 8;                // pieces per wave and stage
    constexpr int KB1 = GM::KB1, NK1 = C / KB1;                // phase-1 ring stages per unit
    static_assert(PC1 % 8 == 0 && PC3 % 8 == 0 && C % KB1 == 0, "ring pieces");

    char* regA = smem;
    char* regB = smem + GM::AB;
    float* tbl = reinterpret_cast<float*>(smem + GM::AB + GM::QB);       // [G][176]: bias * log2 e (169.. = -1e30)
    char* ytile = regA;                                        // phase 1
    char* otile = regA + 2 * B2_HALF;                          // dO_g [MR][OC]
    char* qtile = regB;                                        // dqkv of the group [MR][QC]

    const int tid0 = threadIdx.x;
    // unit order: blocks b and b + 8 share an XCD (observed round-robin placement; speed only), so the blocks of one XCD take a
    // contiguous run of units: the head groups of a window pair then share an L2 (its dY rows are fetched once, not NHG times)
    const int gpx = gridDim.x / 8;
    const int bid = ((int)blockIdx.x < gpx * 8 && gpx % NHG == 0) ? ((int)blockIdx.x & 7) * gpx + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int hg = bid % NHG;                                   // (grid is a multiple of NHG: the head group of a workgroup is fixed)
    const int head0 = hg * G;
    const float sc2 = p.scale * LOG2E_;
    const float MASKL2 = -100.0f * LOG2E_;
    const int nwx = p.res / 7;

    // once per workgroup: bias values of the group's heads (times log2 e), the pair facts of this thread's score elements
    for (int i = tid0; i < G * 176; i += 512) {
        const int h = i / 176, e = i - h * 176;
        tbl[i] = e < 169 ? p.bias_table[e * p.nH + head0 + h] * LOG2E_ : NEG_BIG;
    }
    uint32_t ridx[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) ridx[t] = SWIN_PAIRS.ridx[tid0 & 255][t];
    uint32_t rowA = 0, colA = 0;
    if (p.shift) { const uint32_t b = SWIN_PAIRS.bits3[tid0 & 255]; rowA = b & 0xffffu; colA = b >> 16; }          // shift == 3 (host check)
    // relative-position-bias gradient: thread ht < 169 of a half owns table entry ht = (dy + 6) * 13 + (dx + 6) and adds up its
    // diagonal of every dS image the half produces (bf16 values, f32 sums: one register per head instead of 16 per lane)
    float dbsum[G];
#pragma unroll
    for (int i = 0; i < G; ++i) dbsum[i] = 0.f;
    // entry (dy, dx) sums dS[q][q - (7 dy + dx)] over the q = (qy, qx) with qy - dy and qx - dx inside the window: ONE element offset
    // per thread and 7 + 7 validity flags (hipcc keeps them as lane masks in scalar registers: an s_and + a v_cndmask per element).
    // The 169 entries are dealt to the four waves of a half by the query rows they can touch, so that no wave reads more than four
    // of the seven rows: wave 0: dy = -6 .. -3 (entries 0 .. 51, rows 0 .. 3), wave 1: dy = 3 .. 6 (117 .. 168, rows 3 .. 6),
    // waves 2 / 3: dy = -2 .. 2 (52 .. 115; rows 0 .. 3 / 4 .. 6); the 65th entry of that band (116) rides on lane 52 of waves 0
    // (rows 0 .. 3) and 1 (rows 4 .. 6).  An entry's parts meet when the workgroup stores its sums.
    const int db_w = (tid0 >> 6) & 3, db_l = tid0 & 63;
    const int db_e = db_w == 0 ? (db_l < 52 ? db_l : 116) : (db_w == 1 ? (db_l < 52 ? 117 + db_l : 116) : 52 + db_l);
    const bool db_act = db_w >= 2 || db_l <= 52;
    const int db_dy = db_e / 13 - 6, db_dx = db_e % 13 - 6;
    const int db_off = -(db_dy * 7 + db_dx);
    const int db_r0 = (db_w == 0 || db_w == 2) ? 0 : ((db_w == 1 && db_l < 52) ? 3 : 4);      // first row of this thread's part
    const int db_r1 = (db_w == 0 || db_w == 2) ? 3 : 6;                                        // last
    bool db_rok[7], db_cok[7];
#pragma unroll
    for (int a = 0; a < 7; ++a) {
        db_rok[a] = db_act && a >= db_r0 && a <= db_r1 && a - db_dy >= 0 && a - db_dy <= 6;
        db_cok[a] = a - db_dx >= 0 && a - db_dx <= 6;
    }

#pragma unroll 1
    for (int unit = bid; unit < p.nunits; unit += gridDim.x) {
        // every per-lane quantity is derived INSIDE the loop from an opaque copy of the thread id (wmsa2.hip: with one unit per
        // workgroup loop-invariant hoisting buys nothing and costs dozens of registers carried through every phase)
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, c15 = lane & 15;
        const int half = wave >> 2, hw = wave & 3, ht = tid & 255;
        const int set = unit / NHG;
        const int win0 = 2 * set;
        const long grow0 = (long)win0 * 49;
        __syncthreads();                                       // the previous unit has finished with every region
        WB2_STAMP(0);
        // phase-1 weight ring: stage kk % NST1 <- rows KB1 kk .. of Wproj, columns 32 head0 .. + OC.  ALL slots are requested before
        // anything else of the unit: they are the oldest requests in flight and land with the dY rows
        const T* wsrc = p.wproj + head0 * 32;
        auto fill1 = [&](int kk) {
#pragma unroll
            for (int j = 0; j < PW1; ++j)
                ring_piece<GM::LD1>(wsrc + (long)kk * KB1 * C, C, OC, wave * PW1 + j, lane, lds_addr(regB) + (kk % GM::NST1) * GM::ST1);
        };
#pragma unroll
        for (int kk = 0; kk < GM::NST1; ++kk) if (kk < NK1) fill1(kk);
        asm volatile("" ::: "memory");
        // operands of the attention phase: this half's window, head i -- requested one problem ahead (problem 0: before phase 1)
        const int seq = win0 + half;
        const long rs = (long)seq * 49;
        const int srow = min(ht >> 2, 48), sch = (ht & 3) * 8;
        bf16x8 gq, gk, fv[4];
        float lse_pre = 0.f;
        auto issue = [&](int i) {
            const int h = head0 + min(i, G - 1);
            const T* src = p.qkv + (rs + srow) * 3 * C + h * 32 + sch;
            gq = *reinterpret_cast<const bf16x8*>(src);
            gk = *reinterpret_cast<const bf16x8*>(src + C);
            lse_pre = p.lse[((long)seq * p.nH + h) * 49 + min(ht, 48)];
        };
        auto issue_v = [&](int i) {
            const int h = head0 + min(i, G - 1);
            const T* base = p.qkv + rs * 3 * C + 2 * C + h * 32 + g * 8;
#pragma unroll
            for (int t = 0; t < 4; ++t) fv[t] = *reinterpret_cast<const bf16x8*>(base + (long)min(16 * t + c15, 48) * 3 * C);
        };
        issue(0); issue_v(0);

        // ---------------------------------------------------------------- phase 1: dO_g = dY Wproj[:, group]
        {
            // dY rows of the pair: one contiguous [98][C] block -> registers -> swizzled tile (pad rows: zeros)
            constexpr int CPR = C / 8, NCH = MR * CPR, PER = (NCH + 511) / 512;
            const T* ysrc = p.dy + grow0 * C;
            bf16x8 yv[PER];
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int f = tid + 512 * i;
                yv[i] = f < 98 * CPR ? *reinterpret_cast<const bf16x8*>(ysrc + (long)f * 8) : zero_vec<T>();
            }
            // (the yv loads are YOUNGER than the first ring requests: the compiler's wait for them covers those stages too)
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int f = tid + 512 * i;
                if (f < NCH) *reinterpret_cast<bf16x8*>(ytile + xoff<C>(f / CPR, f % CPR)) = yv[i];
            }
            WB2_STAMP(1);                                      // dY tile written (its loads have arrived)
            // waves 0 .. 5 own one 16-column tile of dO_g each (7 row tiles); waves 6, 7 only keep the barriers
            f32x4 acc[GM::MT];
#pragma unroll
            for (int mt = 0; mt < GM::MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int kk = 0; kk < NK1; ++kk) {
                // younger stages of THIS wave that may still be in flight (PW1 pieces each): all NST1 slots were requested up front,
                // step kk >= 1 refills the slot of stage kk - 1
                const int ahead = min(kk == 0 ? GM::NST1 - 1 : GM::NST1 - 2, NK1 - 1 - kk);
                if (ahead >= 3) wait_vm<3 * PW1>(); else if (ahead == 2) wait_vm<2 * PW1>(); else if (ahead == 1) wait_vm<PW1>(); else wait_vm<0>();
                __syncthreads();
                if (kk >= 1 && kk + GM::NST1 - 1 < NK1) fill1(kk + GM::NST1 - 1);
                if (wave < OC / 16) {
#pragma unroll
                    for (int sb = 0; sb < KB1 / 32; ++sb) {
                        const bf16x8 fb = kfrag<GM::LD1>(regB + (kk % GM::NST1) * GM::ST1 + sb * 32 * GM::LD1 * 2, 16 * wave);
#pragma unroll
                        for (int mt = 0; mt < GM::MT; ++mt) {
                            const bf16x8 fa = *reinterpret_cast<const bf16x8*>(ytile + xoff<C>(16 * mt + c15, 4 * (kk * (KB1 / 32) + sb) + g));
                            M::mma(acc[mt], fb, fa);        // acc[r] <-> (j = 16 wave + 4 g + r, row = 16 mt + c15)
                        }
                    }
                }
            }
            __syncthreads();                                   // every wave is done reading the dY tile: region A changes hands
            WB2_STAMP(2);                                      // phase-1 products done
            if (wave < OC / 16) {
#pragma unroll
                for (int mt = 0; mt < GM::MT; ++mt) {
                    bf16x4 r;
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[e] = (bf16_t)acc[mt][e];
                    const int row = 16 * mt + c15, col = 16 * wave + 4 * g;
                    *reinterpret_cast<bf16x4*>(otile + xoff<OC>(row, col >> 3) + (col & 7) * 2) = r;
                }
            }
            // pad rows of the dqkv tile: zeros (phase 3 multiplies them; the ring that lived here is dead)
            for (int f = tid; f < (MR - 98) * (QC / 8); f += 512) {
                const int row = 98 + f / (QC / 8), ch = f % (QC / 8);
                *reinterpret_cast<bf16x8*>(qtile + xoff<QC>(row, ch)) = zero_vec<T>();
            }
        }

        // phase-3 ring: stage kk <- the 32 rows of Wqkv of (part, head i) = (kk / G, kk % G), all C columns
        auto fill3 = [&](int kk) {
            const int part = kk / G, i = kk - part * G, sl = kk % GM::NST3;
            const unsigned dst = lds_addr(regA) + (sl == 0 ? GM::SLOT0 : (sl - 1) * GM::ST3);
            const T* src = p.wqkv + (long)(part * C + (head0 + i) * 32) * C;
#pragma unroll
            for (int j = 0; j < PW3; ++j) ring_piece<GM::LD3>(src, C, C, wave * PW3 + j, lane, dst);
        };
        // ---------------------------------------------------------------- phase 2: attention backward, window `half`, heads 0 .. G - 1
        {
            char* hb = regA + half * B2_HALF;
            T* qi = reinterpret_cast<T*>(hb);
            T* ki = reinterpret_cast<T*>(hb + B2_IMG);
            T* di = reinterpret_cast<T*>(hb + 2 * B2_IMG);
            T* pi = reinterpret_cast<T*>(hb + 3 * B2_IMG);
            T* si = reinterpret_cast<T*>(hb + 3 * B2_IMG + B2_PIMG);
            float* lse_s = reinterpret_cast<float*>(hb + 3 * B2_IMG + 2 * B2_PIMG);
            uint32_t mb = 0;
            if (p.shift) { const int w = seq % p.nW; mb = ((w / nwx) == nwx - 1 ? rowA : 0u) | ((w % nwx) == nwx - 1 ? colA : 0u); }
            auto store_rows = [&](int row, bool valid, int tcol, const f32x4& v) {     // 4 bf16 of the group's dqkv tile (HBM: once per unit, below)
                bf16x4 r; r[0] = (bf16_t)v[0]; r[1] = (bf16_t)v[1]; r[2] = (bf16_t)v[2]; r[3] = (bf16_t)v[3];
                if (valid) *reinterpret_cast<bf16x4*>(qtile + xoff<QC>(49 * half + row, tcol >> 3) + (tcol & 7) * 2) = r;
            };
            // (not unrolled: a workgroup runs this code once or a few times, and every copy of it is 5 KB more to fetch through a cold
            // instruction cache)
#pragma unroll 1
            for (int i = 0; i < G; ++i) {
                const int h = head0 + i;
                __syncthreads();                               // phase B of the previous problem has read the images (i = 0: the dO tile is complete)
                {
                    const int row = ht >> 2;
                    const bool ok = row < 49;
                    const bf16x8 z = zero_vec<T>();
                    const bf16x8 gd = *reinterpret_cast<const bf16x8*>(otile + xoff<OC>(49 * half + srow, 4 * i + (ht & 3)));
                    *reinterpret_cast<bf16x8*>(qi + row * B2_LD + sch) = ok ? gq : z;
                    *reinterpret_cast<bf16x8*>(ki + row * B2_LD + sch) = ok ? gk : z;
                    *reinterpret_cast<bf16x8*>(di + row * B2_LD + sch) = ok ? gd : z;
                }
                if (ht < 64) lse_s[ht] = ht < 49 ? lse_pre * LOG2E_ : 0.f;
                __syncthreads();
                WB2_STAMP(3 + 3 * i);                          // problem i staged
                if (i == G - 1) fill3(0);                      // the dO tile has been read for the last time: ring slot 0 fills under this problem
                // ---- phase A: keys on accumulator rows, query tile hw on the columns -> dQ, dBias, P and dS images
                {
                    const int tq = hw;
                    f32x4 sc[4], dp[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) { sc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[t] = sc[t]; }
                    const bf16x8 fq = frag_rowmajor<T>(qi, B2_LD, 16 * tq, 0);
                    const bf16x8 fd = frag_rowmajor<T>(di, B2_LD, 16 * tq, 0);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        M::mma(sc[t], frag_rowmajor<T>(ki, B2_LD, 16 * t, 0), fq);
                        M::mma(dp[t], fv[t], fd);
                    }
                    issue(i + 1);                              // next head in flight (last head: re-read, harmless)
                    issue_v(i + 1);                            // the V registers are free again
                    if (i == 0) WB2_STAMP(18);                 // (diagnostic) phase-A score / dP products issued
                    const int q = 16 * tq + c15;
                    const float lse_q = lse_s[q];
                    const float* tb = tbl + i * 176;
                    float dl = 0.f;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float lg = fmaf(sc[t][j], sc2, tb[(ridx[t] >> (8 * j)) & 255]);
                            if (mb & (1u << (4 * t + j))) lg += MASKL2;
                            const float pr = __builtin_amdgcn_exp2f(lg - lse_q);
                            dl = fmaf(pr, dp[t][j], dl);
                            sc[t][j] = pr;
                        }
                    }
                    dl += __shfl_xor(dl, 16, 64);
                    dl += __shfl_xor(dl, 32, 64);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        store4f(pi + q * B2_LDP + 16 * t + 4 * g, sc[t]);
#pragma unroll
                        for (int j = 0; j < 4; ++j) sc[t][j] = sc[t][j] * (dp[t][j] - dl);
                        store4f(si + q * B2_LDP + 16 * t + 4 * g, sc[t]);
                    }
                    if (i == 0) WB2_STAMP(19);                 // (diagnostic) softmax / dS done, images written
                    f32x4 dq[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        const bf16x8 fs = frag_acc<4>(sc, kb, T());
#pragma unroll
                        for (int td = 0; td < 2; ++td) M::mma(dq[td], frag_tok(ki, B2_LD, 16 * td, kb), fs);
                    }
#pragma unroll
                    for (int td = 0; td < 2; ++td) store_rows(q, q < 49, 32 * i + 16 * td + 4 * g, dq[td] * p.scale);
                }
                __syncthreads();                               // P and dS of every query tile are in LDS
                WB2_STAMP(4 + 3 * i);                          // phase A done
                // ---- phase B: key tile hw: dK^T[d, key] = sum_q Q[q, d] dS[q, key], dV^T[d, key] = sum_q dO[q, d] P[q, key]
                {
                    const int tk = hw;
                    f32x4 dk[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, dv[2] = {dk[0], dk[0]};
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        const bf16x8 fs = frag_tok(si, B2_LDP, 16 * tk, kb);
                        const bf16x8 fp = frag_tok(pi, B2_LDP, 16 * tk, kb);
#pragma unroll
                        for (int td = 0; td < 2; ++td) {
                            M::mma(dk[td], frag_tok(qi, B2_LD, 16 * td, kb), fs);
                            M::mma(dv[td], frag_tok(di, B2_LD, 16 * td, kb), fp);
                        }
                    }
                    if (i == 0) WB2_STAMP(16);                 // (diagnostic) phase-B products issued
                    const int k = 16 * tk + c15;
#pragma unroll
                    for (int td = 0; td < 2; ++td) {
                        store_rows(k, k < 49, OC + 32 * i + 16 * td + 4 * g, dk[td] * p.scale);
                        store_rows(k, k < 49, 2 * OC + 32 * i + 16 * td + 4 * g, dv[td]);
                    }
                }
                if (i == 0) WB2_STAMP(17);                     // (diagnostic) phase-B stores issued
                {
                    // <= 28 independent LDS reads at base + 73 q (an invalid element may lie outside the image -- never outside the
                    // workgroup's LDS -- and is dropped by the select); the rows of a wave are wave-uniform
                    const T* sb = si + db_off;
                    float sum = 0.f;
                    const int ra = (hw == 0 || hw == 2) ? 0 : (hw == 1 ? 3 : 4), rb = (hw == 0 || hw == 2) ? 3 : 6;
#pragma unroll
                    for (int qy = 0; qy < 7; ++qy) {
                        if (qy >= ra && qy <= rb) {
#pragma unroll
                            for (int qx = 0; qx < 7; ++qx) {
                                const float v = (float)sb[(7 * qy + qx) * (B2_LDP + 1)];
                                sum += (db_rok[qy] && db_cok[qx]) ? v : 0.f;
                            }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < G; ++k) dbsum[k] += (k == i) ? sum : 0.f;
                }
                WB2_STAMP(5 + 3 * i);                          // phase B done
            }
        }
        // (the ring's requests below are younger than every request of phase 2: its counted waits cover them)
        __syncthreads();                                       // the dqkv tile is complete; region A is free
        WB2_STAMP(12);

        // ---------------------------------------------------------------- phase 3: partial dXn = dqkv_g Wqkv[group rows, :]
        {
#pragma unroll
            for (int kk = 1; kk < GM::NST3 - 1; ++kk) fill3(kk);          // (stage 0 is on its way since the last problem was staged)
            f32x4 acc[CT3][GM::MT];
#pragma unroll
            for (int ct = 0; ct < CT3; ++ct)
#pragma unroll
                for (int mt = 0; mt < GM::MT; ++mt) acc[ct][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int kk = 0; kk < KS3; ++kk) {
                if (kk + GM::NST3 - 2 < KS3) wait_vm<(GM::NST3 - 2) * PW3>(); else wait_vm<0>();
                __syncthreads();
                if (kk + GM::NST3 - 1 < KS3) fill3(kk + GM::NST3 - 1);
                const char* st = regA + (kk % GM::NST3 == 0 ? GM::SLOT0 : (kk % GM::NST3 - 1) * GM::ST3);
                bf16x8 fa[GM::MT];
#pragma unroll
                for (int mt = 0; mt < GM::MT; ++mt) fa[mt] = *reinterpret_cast<const bf16x8*>(qtile + xoff<QC>(16 * mt + c15, 4 * kk + g));
#pragma unroll
                for (int ct = 0; ct < CT3; ++ct) {
                    const int nt = wave + 8 * ct;
                    if (nt < NT3) {
                        const bf16x8 fb = kfrag<GM::LD3>(st, 16 * nt);
#pragma unroll
                        for (int mt = 0; mt < GM::MT; ++mt) M::mma(acc[ct][mt], fb, fa[mt]);     // acc[r] <-> (n = 16 nt + 4 g + r, row = 16 mt + c15)
                    }
                }
            }
            WB2_STAMP(13);                                     // phase-3 products done
            // partial rows through LDS: the accumulators hold 4 columns of 16 rows per register quad (8-byte pieces of 16 rows per
            // store), the tile gives every thread whole 16-byte chunks of consecutive rows
            __syncthreads();                                   // every wave is done with the ring
#pragma unroll
            for (int ct = 0; ct < CT3; ++ct) {
                const int nt = wave + 8 * ct;
                if (nt < NT3) {
#pragma unroll
                    for (int mt = 0; mt < GM::MT; ++mt) {
                        bf16x4 r;
#pragma unroll
                        for (int e = 0; e < 4; ++e) r[e] = (bf16_t)acc[ct][mt][e];
                        const int col = 16 * nt + 4 * g;
                        *reinterpret_cast<bf16x4*>(regA + xoff<C>(16 * mt + c15, col >> 3) + (col & 7) * 2) = r;
                    }
                }
            }
            __syncthreads();
            {
                constexpr int CPR = C / 8;
                T* dst = p.dxn + (long)hg * p.part_stride + grow0 * C;
                for (int f = tid; f < 98 * CPR; f += 512)
                    *reinterpret_cast<bf16x8*>(dst + (long)f * 8) = *reinterpret_cast<const bf16x8*>(regA + xoff<C>(f / CPR, f % CPR));
                // dq / dk / dv of the group (the qkv weight gradient reads them): the tile is still whole -- 16-byte pieces, three
                // 192-byte runs per row (q, k, v columns of the group's heads)
                T* dq0 = p.dqkv + grow0 * 3 * C + head0 * 32;
                for (int f = tid; f < 98 * (QC / 8); f += 512) {
                    const int row = f / (QC / 8), c = f - row * (QC / 8), part = c / (OC / 8), cc = c - part * (OC / 8);
                    *reinterpret_cast<bf16x8*>(dq0 + (long)row * 3 * C + part * C + cc * 8) = *reinterpret_cast<const bf16x8*>(qtile + xoff<QC>(row, c));
                }
            }
        }
    }

    { const int unit = bid; WB2_STAMP(14); }
    // ---- relative-position-bias gradient of the workgroup's units: the (up to four) parts of an entry -- two row ranges x two halves
    // -- are added in LDS in a fixed order, then the workgroup stores its [G][169] sums (no atomics: mvlt_swin_wmsa2_bwd_dbias adds
    // the workgroups up in a fixed order too)
    if (p.dbias_ws) {
        float* sh = reinterpret_cast<float*>(regA);
        __syncthreads();
        for (int i = tid0; i < G * 176; i += 512) sh[i] = 0.f;
        const int rank = (tid0 >> 8) * 2 + (db_w & 1);          // (an entry has at most one contributor per rank)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            __syncthreads();
            if (rank == r && db_act) {
#pragma unroll
                for (int i = 0; i < G; ++i) sh[i * 176 + db_e] += dbsum[i];
            }
        }
        __syncthreads();
        if (tid0 < 169) {
#pragma unroll
            for (int i = 0; i < G; ++i) p.dbias_ws[(long)bid * (G * 169) + i * 169 + tid0] = sh[i * 176 + tid0];
        }
    }
    { const int unit = bid; WB2_STAMP(15); }
}

// table[e][h] += sum over the workgroups wg of head group h / 3 (wg % nhg == h / 3) of ws[wg][(h % 3) * 169 + e]
struct DbiasBatch { int n; MvltSwinDbiasItem it[32]; };
__global__ __launch_bounds__(256) void wb2_dbias_reduce_kernel(const DbiasBatch b) {
    const MvltSwinDbiasItem it = b.it[blockIdx.y];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 169 * it.nH) return;
    const int e = idx / it.nH, h = idx - e * it.nH;
    const int nhg = it.nH / 3, hg = h / 3;
    const float* src = it.ws + (h - 3 * hg) * 169 + e;
    float s0 = 0.f, s1 = 0.f;
    int wg = hg;
    for (; wg + nhg < it.nwg; wg += 2 * nhg) { s0 += src[(long)wg * 507]; s1 += src[(long)(wg + nhg) * 507]; }
    if (wg < it.nwg) s0 += src[(long)wg * 507];
    it.dbias_table[idx] += s0 + s1;
}

thread_local hipEvent_t t_wb2_stop = nullptr;

template <int C, int G>
int launch_wb2(Wb2Dev d, hipStream_t s) {
    using GM = Wb2Geom<C, G>;
    d.nunits = d.nwin / 2 * GM::NHG;
    auto k = wmsa2_bwd_kernel<C, G>;
    static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, GM::bytes) == hipSuccess;
    if (!ok) return MVLT_ERR_LAUNCH;
    static const int ncu = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
    int grid = d.nunits < ncu ? d.nunits : ncu / GM::NHG * GM::NHG;      // a multiple of the group count: a workgroup keeps its head group
    if (grid < GM::NHG) return MVLT_ERR_UNSUPPORTED;
    if (t_wb2_stop) { hipExtLaunchKernelGGL(k, dim3(grid), dim3(512), GM::bytes, s, nullptr, t_wb2_stop, 0, d); t_wb2_stop = nullptr; }
    else hipLaunchKernelGGL(k, dim3(grid), dim3(512), GM::bytes, s, d);
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}

int run_wb2(const MvltSwinWmsa* p, hipStream_t s) {
    MVLT_CHECK(p && p->dy_win && p->qkv_win && p->lse && p->wproj && p->wqkv && p->bias_table && p->dqkv && p->dxn_win, MVLT_ERR_ARG);
    MVLT_CHECK(p->B > 0 && p->res > 0 && p->res % 7 == 0 && p->nH > 0, MVLT_ERR_ARG);
    MVLT_CHECK(aligned16(p->dy_win) && aligned16(p->qkv_win) && aligned16(p->wproj) && aligned16(p->wqkv) && aligned16(p->dqkv) && aligned16(p->dxn_win), MVLT_ERR_ARG);
    if (!mvlt_swin_wmsa2_bwd_parts(p->dtype, p->B, p->res, p->C, p->nH) || !(p->shift == 0 || p->shift == 3)) return MVLT_ERR_UNSUPPORTED;
    const int nW = (p->res / 7) * (p->res / 7);
    Wb2Dev d{};
    d.nwin = p->B * nW; d.nW = nW; d.res = p->res; d.shift = p->shift; d.nH = p->nH;
    if ((double)d.nwin * 49 * 3 * p->C * 2 >= 2147483648.0) return MVLT_ERR_UNSUPPORTED;          // buffer-store range check: < 2 GB
    d.dy = reinterpret_cast<const T*>(p->dy_win); d.qkv = reinterpret_cast<const T*>(p->qkv_win); d.lse = p->lse;
    d.wproj = reinterpret_cast<const T*>(p->wproj); d.wqkv = reinterpret_cast<const T*>(p->wqkv);
    d.bias_table = p->bias_table; d.scale = p->scale;
    d.dqkv = reinterpret_cast<T*>(p->dqkv); d.dxn = reinterpret_cast<T*>(p->dxn_win);
    d.part_stride = (long)d.nwin * 49 * p->C;
    d.dbias_ws = p->dbias_ws;
    if (p->C == 384) return launch_wb2<384, 3>(d, s);
    if (p->C == 192) return launch_wb2<192, 3>(d, s);
    if (p->C == 96) return launch_wb2<96, 3>(d, s);
    return MVLT_ERR_UNSUPPORTED;
}

}  // namespace

// grid of the launch for this shape (= rows of the dbias workspace), 0 = shape not covered
extern "C" int mvlt_swin_wmsa2_bwd_workgroups(int dtype, int B, int res, int C, int nH) {
    const int nhg = mvlt_swin_wmsa2_bwd_parts(dtype, B, res, C, nH);
    if (!nhg) return 0;
    const int nunits = B * (res / 7) * (res / 7) / 2 * nhg;
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    return nunits < ncu ? nunits : ncu / nhg * nhg;
}

extern "C" int mvlt_swin_wmsa2_bwd_dbias(const MvltSwinDbiasItem* items, int n, void* stream) {
    MVLT_CHECK(items && n > 0, MVLT_ERR_ARG);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    for (int i0 = 0; i0 < n; i0 += 32) {
        DbiasBatch b;
        b.n = n - i0 < 32 ? n - i0 : 32;
        int maxh = 0;
        for (int i = 0; i < b.n; ++i) {
            b.it[i] = items[i0 + i];
            MVLT_CHECK(b.it[i].ws && b.it[i].dbias_table && b.it[i].nwg > 0 && b.it[i].nH > 0 && b.it[i].nH % 3 == 0 && b.it[i].nwg % (b.it[i].nH / 3) == 0, MVLT_ERR_ARG);
            if (b.it[i].nH > maxh) maxh = b.it[i].nH;
        }
        hipLaunchKernelGGL(wb2_dbias_reduce_kernel, dim3(ceil_div(169 * maxh, 256), b.n), dim3(256), 0, s, b);
        MVLT_LAUNCH_CHECK();
    }
    return MVLT_OK;
}

// number of partial dXn tensors the one-launch backward writes for this shape (nH / 3); 0 = shape not covered
extern "C" int mvlt_swin_wmsa2_bwd_parts(int dtype, int B, int res, int C, int nH) {
    if (dtype != MVLT_BF16 || nH * 32 != C || res % 7 || nH % 3) return 0;
    if ((B * (res / 7) * (res / 7)) % 2) return 0;
    return (C == 384 || C == 192 || C == 96) ? nH / 3 : 0;
}

#ifdef WB2_TRACE
extern "C" int mvlt_swin_wmsa2_bwd_trace_buffer(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_wb2_trace), &buf, sizeof(buf)) == hipSuccess ? MVLT_OK : MVLT_ERR_LAUNCH;
}
#endif

extern "C" int mvlt_swin_wmsa2_bwd(const MvltSwinWmsa* p, void* stream) { return run_wb2(p, reinterpret_cast<hipStream_t>(stream)); }

extern "C" int mvlt_swin_wmsa2_bwd_ev(const MvltSwinWmsa* p, void* stream, void* event) {
    MVLT_CHECK(event, MVLT_ERR_ARG);
    t_wb2_stop = reinterpret_cast<hipEvent_t>(event);
    const int rc = run_wb2(p, reinterpret_cast<hipStream_t>(stream));
    if (t_wb2_stop) { t_wb2_stop = nullptr; }                  // no kernel took it (an error return): nothing is recorded
    return rc;
}
