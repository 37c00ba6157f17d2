// HBM-bound data-movement / elementwise / loss / optimizer kernels for gfx950.
// Every kernel moves 8-16 bytes per lane per access, coalesced, and is
// launched with enough workgroups to cover the 256 CUs (grid-stride beyond).
#include "common.h"

namespace {

inline int grid_for(long n_items, int per_block, int cap = 8192) {
    long b = (n_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

// ------------------------------------------------------------------ im2col (PatchEmbed)
template <typename T>
__global__ __launch_bounds__(256) void im2col_kernel(const float* img, T* cols, int B, int Cin, int S, int P) {
    // one thread = one (token, c, dy): P contiguous pixels -> P contiguous cols
    const int G = S / P, K = Cin * P * P;
    const long total = (long)B * G * G * Cin * P;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int dy = (int)(idx % P);
        const int c = (int)((idx / P) % Cin);
        const long tok = idx / ((long)P * Cin);
        const int pw = (int)(tok % G), ph = (int)((tok / G) % G), b = (int)(tok / ((long)G * G));
        const float* src = img + (((long)b * Cin + c) * S + ph * P + dy) * S + pw * P;
        T* dst = cols + tok * K + (c * P + dy) * P;
        if (P == 4) { store4f(dst, load4f(src)); }
        else for (int dx = 0; dx < P; ++dx) dst[dx] = from_f<T>(src[dx]);
    }
}

// ------------------------------------------------------------------ VL embeddings
struct EmbDev {
    int B, n_img, T, H, L;
    const int64_t* text; const void* img; const float* word; const float* pos; const float* type;
    int cls_id, sep_id, pos_offset, type_override;
    void* out; const void* dout; void* dimage; float* dword; float* dpos; float* dtype_emb;
    const int* row_start; const int* seq_len; const int* pos_offset_dev;
    int pos_rows, type_rows;
};
// row of (b, posi) in the activation matrix, -1 when the position is not materialised (packed layout)
MVLT_DEV long emb_row(const EmbDev& p, int b, int posi) {
    if (!p.row_start) return (long)b * p.L + posi;
    return posi < p.seq_len[b] ? (long)p.row_start[b] + posi : -1;
}

MVLT_DEV int emb_word_id(const EmbDev& p, int b, int posi) {
    // returns word id, or -1 when the source is the image feature
    if (p.n_img < 0) return (int)p.text[(long)b * p.T + posi];            // cached step: text only
    if (posi == 0) return p.cls_id;
    if (posi <= p.n_img) return -1;
    if (posi == p.n_img + 1) return p.sep_id;
    return (int)p.text[(long)b * p.T + (posi - p.n_img - 2)];
}

// Packing plan of a ragged caption batch (mvlt_pack_plan): one workgroup, no host involvement.
__global__ __launch_bounds__(256) void pack_plan_kernel(const int64_t* text, const int64_t* labels, int B, int T, int n_img,
                                                        int* row_start, int* seq_len, int* total, int64_t* row_start64,
                                                        int64_t* text_row) {
    // caption length = 1 + last position with a non-zero id or a label: all B*T positions are looked at independently
    // (a backwards scan per sample is a chain of dependent global loads: 25 us for T = 80) and meet in seq_len by atomicMax
    __shared__ int slen[1024];
    const bool in_lds = B <= 1024;                  // (LDS atomics: a global atomicMax per position serialises per sample)
    int* lens = in_lds ? slen : seq_len;
    for (int b = threadIdx.x; b < B; b += 256) lens[b] = 0;
    __syncthreads();
    for (long i = threadIdx.x; i < (long)B * T; i += 256) {
        const int b = (int)(i / T), t = (int)(i - (long)b * T);
        if (text[i] != 0 || (labels && labels[i] >= 0)) atomicMax(&lens[b], t + 1);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < B; b += 256) seq_len[b] = lens[b] + n_img + 2;
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int b = 0; b < B; ++b) { row_start[b] = acc; if (row_start64) row_start64[b] = acc; acc += seq_len[b]; }
        total[0] = acc;
    }
    __syncthreads();
    if (text_row) {
        for (long i = threadIdx.x; i < (long)B * T; i += 256) {
            const int b = (int)(i / T), t = (int)(i - (long)b * T);
            const int len = seq_len[b] - (n_img + 2);
            text_row[i] = t < len ? (int64_t)row_start[b] + n_img + 2 + t : (int64_t)row_start[b];
        }
    }
}

// Labelled rows first (mvlt_label_plan): stable partition of the N caption positions by (label >= 0), one workgroup.
__global__ __launch_bounds__(1024) void label_plan_kernel(const int64_t* labels, const int64_t* text_row, int N,
                                                          int* gather_row, int64_t* sel_labels, int* count) {
    __shared__ int part[1024];
    __shared__ int total;
    const int per = (N + 1023) / 1024;
    const int lo = min(threadIdx.x * per, N), hi = min(lo + per, N);
    int c = 0;
    for (int i = lo; i < hi; ++i) c += labels[i] >= 0;
    part[threadIdx.x] = c;
    __syncthreads();
    if (threadIdx.x == 0) {                       // 1024 partial counts: a serial scan is ~1 us
        int acc = 0;
        for (int t = 0; t < 1024; ++t) { const int v = part[t]; part[t] = acc; acc += v; }
        total = acc;
        count[0] = acc;
    }
    __syncthreads();
    int il = part[threadIdx.x];                   // labelled positions before this thread's segment
    int iu = total + (lo - il);                   // unlabelled ones go behind all labelled rows, in order
    for (int i = lo; i < hi; ++i) {
        const int64_t lab = labels[i];
        const int slot = lab >= 0 ? il++ : iu++;
        gather_row[slot] = text_row ? (int)text_row[i] : i;
        sel_labels[slot] = lab;
    }
}

// out[rowmap[i], :] = in[i, :] for i < *count (unique destination rows: the backward pass of the labelled-row gather)
template <typename T>
__global__ __launch_bounds__(256) void rows_scatter_kernel(const T* in, T* out, int rows, int C, const int* rowmap, const int* count) {
    const int n = min(rows, *count);
    const int CV = C / 4;
    const long total = (long)n * CV;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % CV) * 4, i = (int)(idx / CV);
        store4f(out + (long)rowmap[i] * C + c, load4f(in + (long)i * C + c));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const EmbDev p) {
    const int HV = p.H / 4;
    const long total = (long)p.B * p.L * HV;
    const T* img = reinterpret_cast<const T*>(p.img);
    T* out = reinterpret_cast<T*>(p.out);
    const int pos_off = p.pos_offset + (p.pos_offset_dev ? *p.pos_offset_dev : 0);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % HV) * 4;
        const int posi = (int)((idx / HV) % p.L), b = (int)(idx / ((long)HV * p.L));
        const long row = emb_row(p, b, posi);
        if (row < 0) continue;
        const int wid = emb_word_id(p, b, posi);
        f32x4 v = wid >= 0 ? load4f(p.word + (long)wid * p.H + c)
                           : load4f(img + ((long)b * p.n_img + (posi - 1)) * p.H + c);
        const int ty = p.type_override >= 0 ? p.type_override : (posi <= p.n_img + 1 ? 1 : 0);
        v += load4f(p.type + (long)ty * p.H + c);
        v += load4f(p.pos + (long)(posi + pos_off) * p.H + c);
        store4f(out + row * p.H + c, v);
    }
}

// dimage copy + scatter-add of word rows (f32 atomics, one contiguous row per wave-instruction group)
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_tokens_kernel(const EmbDev p) {
    const int HV = p.H / 4;
    const long total = (long)p.B * p.L * HV;
    const T* dout = reinterpret_cast<const T*>(p.dout);
    T* dimg = reinterpret_cast<T*>(p.dimage);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % HV) * 4;
        const int posi = (int)((idx / HV) % p.L), b = (int)(idx / ((long)HV * p.L));
        const long row = emb_row(p, b, posi);
        if (row < 0) continue;
        const f32x4 g = load4f(dout + row * p.H + c);
        const int wid = emb_word_id(p, b, posi);
        if (wid < 0) { if (dimg) store4f(dimg + ((long)b * p.n_img + (posi - 1)) * p.H + c, g); }
        else if (p.dword && posi != 0 && posi != p.n_img + 1) {
            float* d = p.dword + (long)wid * p.H + c;
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(d + e, g[e]);
        }
    }
}
// Position / token-type table gradients and the [CLS] / [SEP] word rows (model.py:133-158 backward): batch sums in a FIXED
// order, no atomics.  A workgroup owns 32 columns; its 1024 threads are 16 column pairs x 64 position lanes.  Every thread
// sums the batch for its positions (B independent loads in flight), OVERWRITES the position row, and keeps the type sums of
// its positions; the 64 lanes then meet in LDS and are added in lane order.  With the table sizes given, every row that
// receives no gradient is zeroed here as well (the caller clears nothing but the word table).
constexpr int EBP_COLS = 32, EBP_LANES = 64;
template <typename T>
__global__ __launch_bounds__(1024) void embed_bwd_pos_kernel(const EmbDev p) {
    __shared__ float red[2][EBP_LANES][EBP_COLS];
    const int cp = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = blockIdx.x * EBP_COLS + 2 * cp;
    const bool cv = c < p.H;                                  // (H % 4 == 0: a pair never straddles the end)
    const T* dout = reinterpret_cast<const T*>(p.dout);
    float t0x = 0.f, t0y = 0.f, t1x = 0.f, t1y = 0.f;         // type rows 0 / 1 (or the override row in t0)
    for (int posi = pl; posi < p.L; posi += EBP_LANES) {
        float sx = 0.f, sy = 0.f;
        if (cv) {
            for (int b = 0; b < p.B; ++b) {
                const long row = emb_row(p, b, posi);
                if (row >= 0) { sx += to_f(dout[row * p.H + c]); sy += to_f(dout[row * p.H + c + 1]); }
            }
            if (p.dpos) { float* d = p.dpos + (long)(posi + p.pos_offset) * p.H + c; d[0] = sx; d[1] = sy; }
            if (p.dword && p.n_img >= 0 && (posi == 0 || posi == p.n_img + 1)) {
                // (the token kernel, earlier on this stream, skips these two positions: one writer per element)
                float* d = p.dword + (long)(posi == 0 ? p.cls_id : p.sep_id) * p.H + c;
                d[0] += sx; d[1] += sy;
            }
        }
        const bool first = p.type_override >= 0 || posi > p.n_img + 1;      // -> t0: type row 0 (text) or the override row
        if (first) { t0x += sx; t0y += sy; } else { t1x += sx; t1y += sy; }
    }
    // rows of the position table outside [pos_offset, pos_offset + L): no gradient
    if (cv && p.dpos && p.pos_rows > 0)
        for (int r = pl; r < p.pos_rows; r += EBP_LANES)
            if (r < p.pos_offset || r >= p.pos_offset + p.L) { float* d = p.dpos + (long)r * p.H + c; d[0] = 0.f; d[1] = 0.f; }
    if (!p.dtype_emb) return;
    red[0][pl][2 * cp] = t0x; red[0][pl][2 * cp + 1] = t0y;
    red[1][pl][2 * cp] = t1x; red[1][pl][2 * cp + 1] = t1y;
    __syncthreads();
    if (threadIdx.x < 2 * EBP_COLS) {
        const int which = threadIdx.x / EBP_COLS, col = threadIdx.x % EBP_COLS, cc = blockIdx.x * EBP_COLS + col;
        if (cc < p.H) {
            float s = 0.f;
            for (int l = 0; l < EBP_LANES; ++l) s += red[which][l][col];
            const int row0 = p.type_override >= 0 ? p.type_override : 0;       // which == 0
            if (which == 0) p.dtype_emb[(long)row0 * p.H + cc] = s;
            else if (p.type_override < 0) p.dtype_emb[(long)1 * p.H + cc] = s;
            // rows of the type table that are not in use
            if (which == 0)
                for (int r = 0; r < p.type_rows; ++r) {
                    const bool used = p.type_override >= 0 ? r == p.type_override : r < 2;
                    if (!used) p.dtype_emb[(long)r * p.H + cc] = 0.f;
                }
        }
    }
}

// ------------------------------------------------------------------ row transform
template <typename T>
__global__ __launch_bounds__(256) void rows_transform_kernel(const T* in, T* out, int rows, int C, const int* rowmap,
                                                            const float* rowscale, int rps, uint32_t thresh,
                                                            float dscale, uint64_t seed, uint32_t tag) {
    const int CV = C / 4;
    const long total = (long)rows * CV;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % CV) * 4, i = (int)(idx / CV);
        const int src = rowmap ? rowmap[i] : i;
        f32x4 v = load4f(in + (long)src * C + c);
        if (thresh) {
            const uint32_t base = (uint32_t)src * (uint32_t)C + c;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = rng_keep(seed, tag, base + e, thresh) ? v[e] * dscale : 0.f;
        }
        if (rowscale) { const float s = rowscale[src / rps];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= s; }
        store4f(out + (long)i * C + c, v);
    }
}

// ------------------------------------------------------------------ elementwise
template <typename S, typename D>
__global__ __launch_bounds__(256) void cast_kernel(const S* src, D* dst, long n) {
    const long nv = n / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256)
        store4f(dst + 4 * i, load4f(src + 4 * i));
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[nv * 4 + threadIdx.x] = from_f<D>(to_f(src[nv * 4 + threadIdx.x]));
}
template <typename T, int OP>   // 0 gelu, 1 tanh, 2 tanh-bwd (y, dy -> dx), 3 gelu-bwd (x, dy -> dx)
__global__ __launch_bounds__(256) void unary_kernel(const T* a, const T* b, T* o, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float x = to_f(a[i]);
        float r;
        if (OP == 0) r = gelu_f(x);
        else if (OP == 1) r = tanhf(x);
        else if (OP == 2) r = to_f(b[i]) * (1.0f - x * x);
        else r = to_f(b[i]) * gelu_grad_f(x);
        o[i] = from_f<T>(r);
    }
}
__global__ void dropout_mask_kernel(uint8_t* keep, long n, uint32_t thresh, uint64_t seed, uint32_t tag) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        keep[i] = rng_keep(seed, tag, (uint32_t)i, thresh) ? 1 : 0;
}
__global__ void droppath_kernel(float* scale, int B, uint32_t thresh, float inv_keep, uint64_t seed, uint32_t tag) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) scale[b] = rng_keep(seed, tag, (uint32_t)b, thresh) ? inv_keep : 0.f;
}
// all the DropPath scales of a forward pass in one launch: row r (one residual branch of one block) drops with probs[r]
__global__ void droppath_rows_kernel(float* scale, const float* probs, int rows, int B, uint64_t seed, uint32_t tag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * B) return;
    const int r = i / B, b = i - r * B;
    const float p = probs[r];
    const uint32_t thresh = (uint32_t)((double)p * 4294967296.0);
    scale[i] = (p <= 0.f || rng_keep(seed, tag + (uint32_t)r, (uint32_t)b, thresh)) ? 1.0f / (1.0f - p) : 0.f;
}

// ------------------------------------------------------------------ column sums
constexpr int COLSUM_ROWS = 128;   // partial rows
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* x, long ld, int M, int N, float* part) {
    // block (bx over 4-column groups x 64, by over row slices): thread = 4 columns x row slice
    const int cg = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rsub = threadIdx.x >> 6;                       // 4 row sub-slices per block
    __shared__ f32x4 red[4][64];
    f32x4 acc{0.f, 0.f, 0.f, 0.f};
    const int c = cg * 4;
    if (c < N) {
        const int rows_per = (M + gridDim.y - 1) / gridDim.y;
        const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
        for (int r = r0 + rsub; r < r1; r += 4) {
            if (c + 3 < N && (ld & 3) == 0) acc += load4f(x + (long)r * ld + c);
            else for (int e = 0; e < 4; ++e) if (c + e < N) acc[e] += to_f(x[(long)r * ld + c + e]);
        }
    }
    red[rsub][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rsub == 0 && c < N) {
        f32x4 s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        for (int e = 0; e < 4; ++e) if (c + e < N) part[(long)blockIdx.y * N + c + e] = s[e];
    }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* part, int nparts, int N, float* out, int accumulate) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < N) for (int i = w; i < nparts; i += 4) s += part[(long)i * N + c];
    red[w][lane] = s;
    __syncthreads();
    if (w == 0 && c < N) {
        s = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
        out[c] = accumulate ? out[c] + s : s;
    }
}

// ------------------------------------------------------------------ cross entropy
template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const T* logits, long ld, int rows, int V, const int64_t* labels,
                                                    float* lse, float* loss_sum, float* count, const int* rows_dev) {
    // one workgroup per row (rows with ignore_index are skipped entirely)
    __shared__ float red[8];
    const int r = blockIdx.x;
    if (rows_dev && r >= *rows_dev) return;            // rows beyond the device-side count do not exist
    const long lab = labels[r];
    if (lab < 0) { if (threadIdx.x == 0 && lse) lse[r] = 0.f; return; }
    const T* x = logits + (long)r * ld;
    float mx = -3.0e38f;
    for (int c = threadIdx.x * 4; c < V; c += 1024) {
        if (c + 3 < V && (ld & 3) == 0) { const f32x4 v = load4f(x + c); mx = fmaxf(fmaxf(mx, fmaxf(v[0], v[1])), fmaxf(v[2], v[3])); }
        else for (int e = 0; e < 4 && c + e < V; ++e) mx = fmaxf(mx, to_f(x[c + e]));
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.f;
    for (int c = threadIdx.x * 4; c < V; c += 1024) {
        if (c + 3 < V && (ld & 3) == 0) { const f32x4 v = load4f(x + c); s += __expf(v[0] - mx) + __expf(v[1] - mx) + __expf(v[2] - mx) + __expf(v[3] - mx); }
        else for (int e = 0; e < 4 && c + e < V; ++e) s += __expf(to_f(x[c + e]) - mx);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float l = mx + logf(red[4] + red[5] + red[6] + red[7]);
        if (lse) lse[r] = l;
        atomicAdd(loss_sum, l - to_f(x[lab]));
        atomicAdd(count, 1.0f);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const T* logits, long ld, int rows, int V, const int64_t* labels,
                                                    const float* lse, const float* count, float gscale,
                                                    const float* gscale_dev, T* dl, const int* rows_dev) {
    const int r = blockIdx.x;
    if (rows_dev && r >= *rows_dev) return;            // never read downstream (MvltGemm.m_dev): not even zero-filled
    if (gscale_dev) gscale *= gscale_dev[0];
    const long lab = labels[r];
    const T* x = logits + (long)r * ld;
    T* d = dl + (long)r * ld;
    const float sc = gscale / fmaxf(count[0], 1.0f);
    const float l = lab >= 0 ? lse[r] : 0.f;
    for (int c = threadIdx.x * 4; c < ld; c += 1024) {
        f32x4 g{0.f, 0.f, 0.f, 0.f};
        if (lab >= 0) {
            if (c + 3 < ld && (ld & 3) == 0) { const f32x4 v = load4f(x + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = (c + e < V) ? sc * (__expf(v[e] - l) - ((long)(c + e) == lab ? 1.f : 0.f)) : 0.f;
            } else for (int e = 0; e < 4 && c + e < V; ++e) g[e] = sc * (__expf(to_f(x[c + e]) - l) - ((long)(c + e) == lab ? 1.f : 0.f));
        }
        if (c + 3 < ld && (ld & 3) == 0) store4f(d + c, g);
        else for (int e = 0; e < 4 && c + e < ld; ++e) d[c + e] = from_f<T>(g[e]);
    }
}

// ------------------------------------------------------------------ AdamW
__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v, bf16_t* sh, long n,
                                                   float lr, float b1, float b2, float eps, float wd,
                                                   float bc1, float bc2_sqrt, float gscale) {
    // torch.optim.AdamW (single tensor path): p *= 1-lr*wd; m,v update; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
    const long nv = n / 4;
    // The fp32 master, both moments and the gradient are read and written once per step (28 of the 30 bytes per
    // parameter): non-temporal accesses, so that the sweep does not push the bf16 shadow (the only thing the next forward
    // reads) and the activations out of the caches: -0.07 ms per step (same-box A/B, 12.78 -> 12.70 ms).
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
        f32x4 pp = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + 4 * i));
        f32x4 gg = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + 4 * i));
        f32x4 mm = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(m + 4 * i));
        f32x4 vv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(v + 4 * i));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = gg[e] * gscale;
            pp[e] *= 1.0f - lr * wd;
            mm[e] = b1 * mm[e] + (1.0f - b1) * gr;
            vv[e] = b2 * vv[e] + (1.0f - b2) * gr * gr;
            pp[e] -= (lr / bc1) * mm[e] / (sqrtf(vv[e]) / bc2_sqrt + eps);
        }
        __builtin_nontemporal_store(pp, reinterpret_cast<f32x4*>(p + 4 * i));
        __builtin_nontemporal_store(mm, reinterpret_cast<f32x4*>(m + 4 * i));
        __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(v + 4 * i));
        if (sh) store4f(sh + 4 * i, pp);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = nv * 4 + threadIdx.x;
        const float gr = g[i] * gscale;
        float pp = p[i] * (1.0f - lr * wd);
        const float mm = b1 * m[i] + (1.0f - b1) * gr, vv = b2 * v[i] + (1.0f - b2) * gr * gr;
        pp -= (lr / bc1) * mm / (sqrtf(vv) / bc2_sqrt + eps);
        p[i] = pp; m[i] = mm; v[i] = vv;
        if (sh) sh[i] = (bf16_t)pp;
    }
}

// ------------------------------------------------------------------ decode: cached attention + argmax
// One wave = one (b, head): all n_new query rows share every K/V load.  Lanes are (key group kg) x (16-byte
// feature chunk fc): a wave instruction reads 64/FC consecutive cache rows of 128 B (1 KB, coalesced).  Keys are
// walked in blocks of KG*8: the 8 K chunks AND the 8 V chunks of a block are all in flight together (one memory
// round trip per block), scores are reduced over the fc lanes by xor shuffles, the softmax is the online
// (running max / sum) form, and the P.V partial sums are reduced over the key groups once at the end.  No LDS,
// no barrier.  (The first version walked the keys one at a time with a dependent 2-byte load each: 68 us per
// layer at past = 200; the two-pass LDS version 21 us.)
// NW waves per (b, head) take the key blocks round-robin (a report of 150 tokens has <= 202 keys = 4 blocks of 64: one
// memory round trip per wave instead of four dependent ones) and meet in LDS: wave 0 merges the (max, sum, P.V) triples.
constexpr int CACHED_MAXNEW = 4, CACHED_MAXK = 1 << 20, CACHED_U = 8;
template <typename T, int NW>
__global__ __launch_bounds__(64 * NW) void attn_cached_kernel(const MvltAttnCached p) {
    constexpr int E = TypeInfo<T>::E, HD = 64, FC = HD / E, KG = 64 / FC;
    using Vec = typename TypeInfo<T>::Vec;
    __shared__ float wm[NW][CACHED_MAXNEW], wl[NW][CACHED_MAXNEW], wo[NW][CACHED_MAXNEW][HD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fc = lane % FC, kg = lane / FC;
    const int h = blockIdx.x % p.nH, b = blockIdx.x / p.nH;
    const int past = p.past_dev ? *p.past_dev : p.past;
    const int C = p.nH * HD;
    const T* qkv = reinterpret_cast<const T*>(p.qkv_new) + (long)b * p.n_new * 3 * C + h * HD + fc * E;
    T* kc = reinterpret_cast<T*>(p.k_cache) + ((long)b * p.nH + h) * p.cache_cap * HD + fc * E;
    T* vc = reinterpret_cast<T*>(p.v_cache) + ((long)b * p.nH + h) * p.cache_cap * HD + fc * E;
    const int nk = past + p.n_new;
    float q[CACHED_MAXNEW][E], o[CACHED_MAXNEW][E], m[CACHED_MAXNEW], l[CACHED_MAXNEW];
#pragma unroll
    for (int r = 0; r < CACHED_MAXNEW; ++r) {
        m[r] = -3.0e38f; l[r] = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) { q[r][e] = 0.f; o[r][e] = 0.f; }
        if (r < p.n_new) {
            const Vec qv = *reinterpret_cast<const Vec*>(qkv + (long)r * 3 * C);
#pragma unroll
            for (int e = 0; e < E; ++e) q[r][e] = to_f(qv[e]) * p.scale;
            if (kg == 0 && wave == 0) {       // append this row's K/V chunk (the loop below reads new rows from qkv_new, not the cache)
                *reinterpret_cast<Vec*>(kc + (long)(past + r) * HD) = *reinterpret_cast<const Vec*>(qkv + (long)r * 3 * C + C);
                *reinterpret_cast<Vec*>(vc + (long)(past + r) * HD) = *reinterpret_cast<const Vec*>(qkv + (long)r * 3 * C + 2 * C);
            }
        }
    }
    for (int k0 = wave * KG * CACHED_U; k0 < nk; k0 += NW * KG * CACHED_U) {
        Vec kv[CACHED_U], vv[CACHED_U];
#pragma unroll
        for (int u = 0; u < CACHED_U; ++u) {
            const int k = k0 + u * KG + kg;
            const int kk = k < nk ? k : 0;                                   // out of range: a valid row, masked below
            const T* ks = kk < past ? kc + (long)kk * HD : qkv + (long)(kk - past) * 3 * C + C;
            const T* vs = kk < past ? vc + (long)kk * HD : qkv + (long)(kk - past) * 3 * C + 2 * C;
            kv[u] = *reinterpret_cast<const Vec*>(ks);
            vv[u] = *reinterpret_cast<const Vec*>(vs);
        }
        float s[CACHED_U][CACHED_MAXNEW];
#pragma unroll
        for (int u = 0; u < CACHED_U; ++u) {
            const int k = k0 + u * KG + kg;
#pragma unroll
            for (int r = 0; r < CACHED_MAXNEW; ++r) {
                float part = 0.f;
#pragma unroll
                for (int e = 0; e < E; ++e) part += q[r][e] * to_f(kv[u][e]);
#pragma unroll
                for (int off = 1; off < FC; off <<= 1) part += __shfl_xor(part, off, 64);
                // causal over the new tokens (model.py:97-104): key k is visible to new row r iff k <= past + r
                s[u][r] = (k < nk && k <= past + r) ? part : -3.0e38f;
            }
        }
#pragma unroll
        for (int r = 0; r < CACHED_MAXNEW; ++r) {
            if (r >= p.n_new) break;
            float mb = s[0][r];
#pragma unroll
            for (int u = 1; u < CACHED_U; ++u) mb = fmaxf(mb, s[u][r]);
#pragma unroll
            for (int off = FC; off < 64; off <<= 1) mb = fmaxf(mb, __shfl_xor(mb, off, 64));
            const float mn = fmaxf(m[r], mb);
            const float alpha = __expf(m[r] - mn);
            m[r] = mn;
            float ls = 0.f;
#pragma unroll
            for (int e = 0; e < E; ++e) o[r][e] *= alpha;
#pragma unroll
            for (int u = 0; u < CACHED_U; ++u) {
                const float pr = __expf(s[u][r] - mn);                     // masked: exp(-3e38 - mn) = 0
                ls += pr;
#pragma unroll
                for (int e = 0; e < E; ++e) o[r][e] += pr * to_f(vv[u][e]);
            }
            l[r] = l[r] * alpha + ls;
        }
    }
#pragma unroll
    for (int r = 0; r < CACHED_MAXNEW; ++r) {
        if (r >= p.n_new) break;
        float lt = l[r];
#pragma unroll
        for (int off = FC; off < 64; off <<= 1) lt += __shfl_xor(lt, off, 64);
        float ov[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            float v = o[r][e];
#pragma unroll
            for (int off = FC; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
            ov[e] = v;
        }
        if (NW == 1) {
            const float inv = 1.0f / lt;
            Vec ovv;
#pragma unroll
            for (int e = 0; e < E; ++e) ovv[e] = from_f<T>(ov[e] * inv);
            if (kg == 0)
                *reinterpret_cast<Vec*>(reinterpret_cast<T*>(p.out) + ((long)b * p.n_new + r) * C + h * HD + fc * E) = ovv;
        } else if (kg == 0) {
            if (fc == 0) { wm[wave][r] = m[r]; wl[wave][r] = lt; }
#pragma unroll
            for (int e = 0; e < E; ++e) wo[wave][r][fc * E + e] = ov[e];
        }
    }
    if (NW > 1) {
        __syncthreads();
        if (wave == 0 && kg == 0) {
#pragma unroll
            for (int r = 0; r < CACHED_MAXNEW; ++r) {
                if (r >= p.n_new) break;
                float mx = wm[0][r];
#pragma unroll
                for (int w = 1; w < NW; ++w) mx = fmaxf(mx, wm[w][r]);
                float lt = 0.f, acc[E];
#pragma unroll
                for (int e = 0; e < E; ++e) acc[e] = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const float a = __expf(wm[w][r] - mx);            // a wave that saw no key: exp(-3e38 - mx) = 0
                    lt += wl[w][r] * a;
#pragma unroll
                    for (int e = 0; e < E; ++e) acc[e] += wo[w][r][fc * E + e] * a;
                }
                const float inv = 1.0f / lt;
                Vec ovv;
#pragma unroll
                for (int e = 0; e < E; ++e) ovv[e] = from_f<T>(acc[e] * inv);
                *reinterpret_cast<Vec*>(reinterpret_cast<T*>(p.out) + ((long)b * p.n_new + r) * C + h * HD + fc * E) = ovv;
            }
        }
    }
}

// fallback (head_dim < 64 or more than 4 new rows): one wave per (b, head, new row), keys serial
template <typename T>
__global__ __launch_bounds__(64) void attn_cached_serial_kernel(const MvltAttnCached p) {
    // one wave = one (b, head, new row); hd = 64 -> lane owns one feature; keys streamed from the cache
    const int lane = threadIdx.x;
    const int past = p.past_dev ? *p.past_dev : p.past;
    const int row = blockIdx.x % p.n_new, h = (blockIdx.x / p.n_new) % p.nH, b = blockIdx.x / (p.n_new * p.nH);
    const int C = p.nH * p.hd;
    const T* qkv = reinterpret_cast<const T*>(p.qkv_new);
    T* kc = reinterpret_cast<T*>(p.k_cache) + ((long)b * p.nH + h) * p.cache_cap * p.hd;
    T* vc = reinterpret_cast<T*>(p.v_cache) + ((long)b * p.nH + h) * p.cache_cap * p.hd;
    // append this row's K/V (each (b,h,row) wave appends its own row)
    const T* src = qkv + ((long)b * p.n_new + row) * 3 * C + h * p.hd;
    if (lane < p.hd) {
        kc[(long)(past + row) * p.hd + lane] = src[C + lane];
        vc[(long)(past + row) * p.hd + lane] = src[2 * C + lane];
    }
    const float q = lane < p.hd ? to_f(src[lane]) * p.scale : 0.f;
    const int nk = past + row + 1;                     // causal over the new tokens (model.py:97-104)
    float m = -3.0e38f, l = 0.f, o = 0.f;
    for (int k = 0; k < nk; ++k) {
        float kv, vv;
        if (k < past) { kv = lane < p.hd ? to_f(kc[(long)k * p.hd + lane]) : 0.f; vv = lane < p.hd ? to_f(vc[(long)k * p.hd + lane]) : 0.f; }
        else {   // rows appended in this launch: read them from qkv_new (other waves may not have stored yet)
            const T* s2 = qkv + ((long)b * p.n_new + (k - past)) * 3 * C + h * p.hd;
            kv = lane < p.hd ? to_f(s2[C + lane]) : 0.f; vv = lane < p.hd ? to_f(s2[2 * C + lane]) : 0.f;
        }
        const float s = wave_sum(q * kv);
        const float mn = fmaxf(m, s);
        const float a = __expf(m - mn), e = __expf(s - mn);
        l = l * a + e; o = o * a + e * vv; m = mn;
    }
    if (lane < p.hd) reinterpret_cast<T*>(p.out)[((long)b * p.n_new + row) * C + h * p.hd + lane] = from_f<T>(o / l);
}

template <typename T>
__global__ __launch_bounds__(256) void argmax_kernel(const T* logits, long ld, int V, int64_t* out) {
    __shared__ float bv[4]; __shared__ int bi[4];
    const T* x = logits + (long)blockIdx.x * ld;
    float best = -3.0e38f; int idx = 0x7fffffff;
    for (int c = threadIdx.x; c < V; c += 256) { const float v = to_f(x[c]); if (v > best) { best = v; idx = c; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(idx, o, 64);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        out[blockIdx.x] = idx;
    }
}

template <typename T>
__global__ __launch_bounds__(64) void softmax_rows_kernel(const T* x, long ld, int V, float* out) {
    const T* r = x + (long)blockIdx.x * ld;
    float mx = -3.0e38f;
    for (int c = threadIdx.x; c < V; c += 64) mx = fmaxf(mx, to_f(r[c]));
    mx = wave_max(mx);
    float s = 0.f;
    for (int c = threadIdx.x; c < V; c += 64) s += __expf(to_f(r[c]) - mx);
    s = wave_sum(s);
    for (int c = threadIdx.x; c < V; c += 64) out[(long)blockIdx.x * V + c] = __expf(to_f(r[c]) - mx) / s;
}

struct ZeroBatch { int n; MvltZeroItem it[32]; };
__global__ __launch_bounds__(256) void zero_batch_kernel(const ZeroBatch b) {
    const MvltZeroItem it = b.it[blockIdx.y];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < it.n; i += (long)gridDim.x * 256) it.ptr[i] = 0.f;
}

// ------------------------------------------------------------------ input pipeline
// one workgroup per (image, channel): exact integer sums -> mean / population variance in f64 -> normalise
__global__ __launch_bounds__(256) void image_normalize_kernel(const uint8_t* hwc, float* chw, int H, int W) {
    __shared__ unsigned long long red[2][4];
    const int c = blockIdx.x, b = blockIdx.y;
    const long n = (long)H * W;
    const uint8_t* src = hwc + (long)b * n * 3 + c;
    unsigned long long s1 = 0, s2 = 0;
    for (long i = threadIdx.x; i < n; i += 256) { const unsigned v = src[i * 3]; s1 += v; s2 += v * v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
    __syncthreads();
    s1 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    s2 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const double mean = (double)s1 / (double)n;
    const double var = (double)s2 / (double)n - mean * mean;
    const float fm = (float)mean, fv = (float)var;
    float* dst = chw + ((long)b * 3 + c) * n;
    for (long i = threadIdx.x; i < n; i += 256) dst[i] = ((float)src[i * 3] - fm) / fv;
}

// one wave per caption
__global__ __launch_bounds__(64) void mlm_mask_kernel(const MvltMlmMask p) {
    __shared__ uint32_t key[1024];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int64_t* in = p.ids_in + (long)b * p.T;
    int64_t* out = p.ids_out + (long)b * p.T;
    int64_t* lab = p.labels + (long)b * p.T;
    for (int t = lane; t < p.T; t += 64) { out[t] = in[t]; lab[t] = -100; }
    if (p.itm_label && p.itm_label[b] == 0) return;
    const int Lf = min(p.full_len[b], 1024);
    if (Lf <= 0) return;
    int n = (int)rint(0.2 * (double)Lf);                 // Python round(): half to even, like rint
    n = min(10, max(1, n));
    for (int i = lane; i < Lf; i += 64) key[i] = rng_u32(p.seed, 2u * (uint32_t)b, (uint32_t)i);
    __syncthreads();
    for (int i = lane; i < Lf; i += 64) {
        const uint32_t k = key[i];
        int rank = 0;
        for (int j = 0; j < Lf; ++j) rank += (key[j] < k) || (key[j] == k && j < i);
        if (rank >= n) continue;                         // the n smallest keys = a uniform n-subset
        int col = i;
        if (Lf > p.T) { if (i == Lf - 1) col = p.T - 1; else if (i >= p.T - 1) continue; }   // cut off by :170-176
        if (col >= p.T) continue;
        const uint32_t r = rng_u32(p.seed, 2u * (uint32_t)b + 1u, (uint32_t)i);
        const float u = (float)(r >> 8) * (1.0f / 16777216.0f);
        lab[col] = in[col];
        if (u < 0.8f) out[col] = p.mask_id;
        else if (u < 0.9f) out[col] = (int64_t)(mix32(r ^ 0x85ebca6bU) % (uint32_t)p.vocab_size);
    }
}

}  // namespace

#define STREAM(s) reinterpret_cast<hipStream_t>(s)
#define BY_DTYPE(dt, CALL_F32, CALL_BF16) do { if ((dt) == MVLT_F32) { CALL_F32; } else if ((dt) == MVLT_BF16) { CALL_BF16; } else return MVLT_ERR_UNSUPPORTED; } while (0)

extern "C" int mvlt_version(void) { return MVLT_ABI_VERSION; }
extern "C" size_t mvlt_sizeof(int struct_id) {
    switch (struct_id) {
        case MVLT_STRUCT_GEMM: return sizeof(MvltGemm);
        case MVLT_STRUCT_LAYERNORM: return sizeof(MvltLayerNorm);
        case MVLT_STRUCT_LAYERNORM_BWD: return sizeof(MvltLayerNormBwd);
        case MVLT_STRUCT_LN_REDUCE_ITEM: return sizeof(MvltLnReduceItem);
        case MVLT_STRUCT_ATTN: return sizeof(MvltAttn);
        case MVLT_STRUCT_SWIN_WMSA: return sizeof(MvltSwinWmsa);
        case MVLT_STRUCT_EMBED: return sizeof(MvltEmbed);
        case MVLT_STRUCT_ATTN_CACHED: return sizeof(MvltAttnCached);
        case MVLT_STRUCT_ZERO_ITEM: return sizeof(MvltZeroItem);
        case MVLT_STRUCT_RANGE: return sizeof(MvltRange);
        case MVLT_STRUCT_MLM_MASK: return sizeof(MvltMlmMask);
        case MVLT_STRUCT_GREEDY_STATE: return sizeof(MvltGreedyState);
        case MVLT_STRUCT_SWIN_DBIAS_ITEM: return sizeof(MvltSwinDbiasItem);
        default: return 0;
    }
}
extern "C" const char* mvlt_arch(void) { return "gfx950"; }

extern "C" int mvlt_im2col_patch(int dtype, const float* img, void* cols, int B, int Cin, int S, int P, void* stream) {
    MVLT_CHECK(img && cols && B > 0 && Cin > 0 && P > 0 && S % P == 0, MVLT_ERR_ARG);
    MVLT_CHECK(aligned16(img) && aligned16(cols), MVLT_ERR_ARG);
    const long total = (long)B * (S / P) * (S / P) * Cin * P;
    const int g = grid_for(total, 256);
    BY_DTYPE(dtype,
             hipLaunchKernelGGL(im2col_kernel<float>, dim3(g), dim3(256), 0, STREAM(stream), img, (float*)cols, B, Cin, S, P),
             hipLaunchKernelGGL(im2col_kernel<bf16_t>, dim3(g), dim3(256), 0, STREAM(stream), img, (bf16_t*)cols, B, Cin, S, P));
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}

static int fill_emb(const MvltEmbed* p, EmbDev& d) {
    MVLT_CHECK(p && p->B > 0 && p->H > 0 && p->H % 4 == 0 && p->word_emb && p->pos_emb && p->type_emb, MVLT_ERR_ARG);
    MVLT_CHECK(p->T == 0 || p->text_ids, MVLT_ERR_ARG);
    d.B = p->B; d.n_img = p->n_img; d.T = p->T; d.H = p->H;
    d.L = p->n_img < 0 ? p->T : p->n_img + 2 + p->T;
    d.text = p->text_ids; d.img = p->image_feature; d.word = p->word_emb; d.pos = p->pos_emb; d.type = p->type_emb;
    d.cls_id = p->cls_id; d.sep_id = p->sep_id; d.pos_offset = p->pos_offset; d.type_override = p->type_override;
    d.out = p->out; d.dout = p->dout; d.dimage = p->dimage; d.dword = p->dword; d.dpos = p->dpos; d.dtype_emb = p->dtype_emb;
    MVLT_CHECK((p->row_start == nullptr) == (p->seq_len == nullptr), MVLT_ERR_ARG);
    d.row_start = p->row_start; d.seq_len = p->seq_len; d.pos_offset_dev = p->pos_offset_dev;
    d.pos_rows = p->pos_rows; d.type_rows = p->type_rows;
    if (p->n_img >= 0) MVLT_CHECK(p->image_feature || p->dout, MVLT_ERR_ARG);
    return MVLT_OK;
}
extern "C" int mvlt_pack_plan(const int64_t* text_ids, const int64_t* labels, int B, int T, int n_img,
                              int32_t* row_start, int32_t* seq_len, int32_t* total_rows, int64_t* row_start64,
                              int64_t* text_row, void* stream) {
    MVLT_CHECK(text_ids && row_start && seq_len && total_rows, MVLT_ERR_ARG);
    MVLT_CHECK(B > 0 && B <= 65536 && T > 0 && n_img >= 0, MVLT_ERR_ARG);
    hipLaunchKernelGGL(pack_plan_kernel, dim3(1), dim3(256), 0, STREAM(stream), text_ids, labels, B, T, n_img, row_start, seq_len,
                       total_rows, row_start64, text_row);
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}

extern "C" int mvlt_label_plan(const int64_t* labels, const int64_t* text_row, int N, int32_t* gather_row,
                               int64_t* sel_labels, int32_t* count, void* stream) {
    MVLT_CHECK(labels && gather_row && sel_labels && count && N > 0 && N <= (1 << 20), MVLT_ERR_ARG);
    hipLaunchKernelGGL(label_plan_kernel, dim3(1), dim3(1024), 0, STREAM(stream), labels, text_row, N, gather_row, sel_labels, count);
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}

extern "C" int mvlt_rows_scatter(int dtype, const void* in, void* out, int rows, int C, const int32_t* rowmap,
                                 const int32_t* count, void* stream) {
    MVLT_CHECK(in && out && rowmap && count && rows > 0 && C > 0 && C % 4 == 0, MVLT_ERR_ARG);
    const int g = grid_for((long)rows * (C / 4), 256);
    BY_DTYPE(dtype, hipLaunchKernelGGL(rows_scatter_kernel<float>, dim3(g), dim3(256), 0, STREAM(stream), (const float*)in, (float*)out, rows, C, rowmap, count),
             hipLaunchKernelGGL(rows_scatter_kernel<bf16_t>, dim3(g), dim3(256), 0, STREAM(stream), (const bf16_t*)in, (bf16_t*)out, rows, C, rowmap, count));
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}

extern "C" int mvlt_embed_fwd(const MvltEmbed* p, void* stream) {
    EmbDev d{}; int rc = fill_emb(p, d); if (rc) return rc;
    MVLT_CHECK(p->out && (p->n_img < 0 || p->image_feature), MVLT_ERR_ARG);
    const int g = grid_for((long)d.B * d.L * (d.H / 4), 256);
    BY_DTYPE(p->dtype, hipLaunchKernelGGL(embed_fwd_kernel<float>, dim3(g), dim3(256), 0, STREAM(stream), d),
             hipLaunchKernelGGL(embed_fwd_kernel<bf16_t>, dim3(g), dim3(256), 0, STREAM(stream), d));
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}
extern "C" int mvlt_embed_bwd(const MvltEmbed* p, void* stream) {
    EmbDev d{}; int rc = fill_emb(p, d); if (rc) return rc;
    MVLT_CHECK(p->dout, MVLT_ERR_ARG);
    const int g = grid_for((long)d.B * d.L * (d.H / 4), 256);
    MVLT_CHECK(p->pos_rows == 0 || p->pos_offset + d.L <= p->pos_rows, MVLT_ERR_ARG);
    MVLT_CHECK(p->type_rows == 0 || (p->type_override >= 0 ? p->type_override < p->type_rows : p->type_rows >= 2), MVLT_ERR_ARG);
    const int g2 = ceil_div(d.H, EBP_COLS);
    BY_DTYPE(p->dtype,
             { hipLaunchKernelGGL(embed_bwd_tokens_kernel<float>, dim3(g), dim3(256), 0, STREAM(stream), d);
               hipLaunchKernelGGL(embed_bwd_pos_kernel<float>, dim3(g2), dim3(1024), 0, STREAM(stream), d); },
             { hipLaunchKernelGGL(embed_bwd_tokens_kernel<bf16_t>, dim3(g), dim3(256), 0, STREAM(stream), d);
               hipLaunchKernelGGL(embed_bwd_pos_kernel<bf16_t>, dim3(g2), dim3(1024), 0, STREAM(stream), d); });
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}

extern "C" int mvlt_rows_transform(int dtype, const void* in, void* out, int rows, int C, const int32_t* rowmap,
                                   const float* rowscale, int rows_per_scale, float dropout_p, uint64_t seed,
                                   uint32_t tag, void* stream) {
    MVLT_CHECK(in && out && rows > 0 && C > 0 && C % 4 == 0 && dropout_p >= 0.f && dropout_p < 1.f, MVLT_ERR_ARG);
    MVLT_CHECK((double)rows * C < 4294967296.0, MVLT_ERR_ARG);
    const uint32_t th = (uint32_t)((double)dropout_p * 4294967296.0);
    const float ds = 1.0f / (1.0f - dropout_p);
    const int rps = rows_per_scale > 0 ? rows_per_scale : 1;
    const int g = grid_for((long)rows * (C / 4), 256);
    BY_DTYPE(dtype,
             hipLaunchKernelGGL(rows_transform_kernel<float>, dim3(g), dim3(256), 0, STREAM(stream), (const float*)in, (float*)out, rows, C, rowmap, rowscale, rps, th, ds, seed, tag),
             hipLaunchKernelGGL(rows_transform_kernel<bf16_t>, dim3(g), dim3(256), 0, STREAM(stream), (const bf16_t*)in, (bf16_t*)out, rows, C, rowmap, rowscale, rps, th, ds, seed, tag));
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}

extern "C" int mvlt_cast(int sd, const void* src, int dd, void* dst, int64_t n, void* stream) {
    MVLT_CHECK(src && dst && n > 0 && aligned16(src) && aligned16(dst), MVLT_ERR_ARG);
    const int g = grid_for(n / 4 + 1, 256);
    hipStream_t s = STREAM(stream);
    if (sd == MVLT_F32 && dd == MVLT_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(g), dim3(256), 0, s, (const float*)src, (bf16_t*)dst, (long)n);
    else if (sd == MVLT_BF16 && dd == MVLT_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(g), dim3(256), 0, s, (const bf16_t*)src, (float*)dst, (long)n);
    else if (sd == MVLT_F32 && dd == MVLT_F32) hipLaunchKernelGGL((cast_kernel<float, float>), dim3(g), dim3(256), 0, s, (const float*)src, (float*)dst, (long)n);
    else if (sd == MVLT_BF16 && dd == MVLT_BF16) hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), dim3(g), dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, (long)n);
    else return MVLT_ERR_UNSUPPORTED;
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}

template <int OP>
static int unary(int dtype, const void* a, const void* b, void* o, int64_t n, void* stream) {
    MVLT_CHECK(a && o && n > 0, MVLT_ERR_ARG);
    const int g = grid_for(n, 256);
    BY_DTYPE(dtype,
             hipLaunchKernelGGL((unary_kernel<float, OP>), dim3(g), dim3(256), 0, STREAM(stream), (const float*)a, (const float*)b, (float*)o, (long)n),
             hipLaunchKernelGGL((unary_kernel<bf16_t, OP>), dim3(g), dim3(256), 0, STREAM(stream), (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)o, (long)n));
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}
extern "C" int mvlt_gelu_fwd(int dtype, const void* x, void* y, int64_t n, void* stream) { return unary<0>(dtype, x, nullptr, y, n, stream); }
extern "C" int mvlt_tanh_fwd(int dtype, const void* x, void* y, int64_t n, void* stream) { return unary<1>(dtype, x, nullptr, y, n, stream); }
extern "C" int mvlt_gelu_bwd(int dtype, const void* x, const void* dy, void* dx, int64_t n, void* stream) {
    MVLT_CHECK(dy, MVLT_ERR_ARG);
    return unary<3>(dtype, x, dy, dx, n, stream);
}
extern "C" int mvlt_tanh_bwd(int dtype, const void* y, const void* dy, void* dx, int64_t n, void* stream) {
    MVLT_CHECK(dy, MVLT_ERR_ARG);
    return unary<2>(dtype, y, dy, dx, n, stream);
}

extern "C" int mvlt_dropout_mask(uint8_t* keep, int64_t n, float p, uint64_t seed, uint32_t tag, void* stream) {
    MVLT_CHECK(keep && n > 0 && n < 4294967296LL && p >= 0.f && p < 1.f, MVLT_ERR_ARG);
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n, 256)), dim3(256), 0, STREAM(stream), keep, (long)n,
                       (uint32_t)((double)p * 4294967296.0), seed, tag);
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}
extern "C" int mvlt_droppath_scale(float* scale, int B, float p, uint64_t seed, uint32_t tag, void* stream) {
    MVLT_CHECK(scale && B > 0 && p >= 0.f && p < 1.f, MVLT_ERR_ARG);
    hipLaunchKernelGGL(droppath_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, STREAM(stream), scale, B,
                       (uint32_t)((double)p * 4294967296.0), 1.0f / (1.0f - p), seed, tag);
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}
extern "C" int mvlt_droppath_scales(float* scale, const float* probs, int rows, int B, uint64_t seed, uint32_t tag, void* stream) {
    MVLT_CHECK(scale && probs && rows > 0 && B > 0 && (long)rows * B < (1L << 30), MVLT_ERR_ARG);
    hipLaunchKernelGGL(droppath_rows_kernel, dim3(ceil_div(rows * B, 256)), dim3(256), 0, STREAM(stream), scale, probs, rows, B, seed, tag);
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}

extern "C" int mvlt_colsum_workspace_rows(int M) { (void)M; return COLSUM_ROWS; }
extern "C" int mvlt_colsum(int dtype, const void* x, int64_t ld, int M, int N, float* out, int accumulate,
                           float* workspace, void* stream) {
    MVLT_CHECK(x && out && workspace && M > 0 && N > 0 && ld >= N, MVLT_ERR_ARG);
    int slices = M / 64; if (slices < 1) slices = 1; if (slices > COLSUM_ROWS) slices = COLSUM_ROWS;
    dim3 grid(ceil_div(ceil_div(N, 4), 64), slices);
    BY_DTYPE(dtype,
             hipLaunchKernelGGL(colsum_partial_kernel<float>, grid, dim3(256), 0, STREAM(stream), (const float*)x, (long)ld, M, N, workspace),
             hipLaunchKernelGGL(colsum_partial_kernel<bf16_t>, grid, dim3(256), 0, STREAM(stream), (const bf16_t*)x, (long)ld, M, N, workspace));
    hipLaunchKernelGGL(colsum_final_kernel, dim3(ceil_div(N, 64)), dim3(256), 0, STREAM(stream), workspace, slices, N, out, accumulate);
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}

extern "C" int mvlt_ce_fwd(int dtype, const void* logits, int64_t ld, int rows, int V, const int64_t* labels,
                           float* lse, float* loss_sum, float* count, void* stream) {
    return mvlt_ce_fwd_ragged(dtype, logits, ld, rows, V, labels, lse, loss_sum, count, nullptr, stream);
}
extern "C" int mvlt_ce_fwd_ragged(int dtype, const void* logits, int64_t ld, int rows, int V, const int64_t* labels,
                                  float* lse, float* loss_sum, float* count, const int32_t* rows_dev, void* stream) {
    MVLT_CHECK(logits && labels && loss_sum && count && rows > 0 && V > 0 && ld >= V, MVLT_ERR_ARG);
    BY_DTYPE(dtype,
             hipLaunchKernelGGL(ce_fwd_kernel<float>, dim3(rows), dim3(256), 0, STREAM(stream), (const float*)logits, (long)ld, rows, V, labels, lse, loss_sum, count, rows_dev),
             hipLaunchKernelGGL(ce_fwd_kernel<bf16_t>, dim3(rows), dim3(256), 0, STREAM(stream), (const bf16_t*)logits, (long)ld, rows, V, labels, lse, loss_sum, count, rows_dev));
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}
extern "C" int mvlt_ce_bwd(int dtype, const void* logits, int64_t ld, int rows, int V, const int64_t* labels,
                           const float* lse, const float* count, float grad_scale, const float* grad_scale_dev,
                           void* dlogits, void* stream) {
    return mvlt_ce_bwd_ragged(dtype, logits, ld, rows, V, labels, lse, count, grad_scale, grad_scale_dev, dlogits, nullptr, stream);
}
extern "C" int mvlt_ce_bwd_ragged(int dtype, const void* logits, int64_t ld, int rows, int V, const int64_t* labels,
                                  const float* lse, const float* count, float grad_scale, const float* grad_scale_dev,
                                  void* dlogits, const int32_t* rows_dev, void* stream) {
    MVLT_CHECK(logits && labels && lse && count && dlogits && rows > 0 && V > 0 && ld >= V, MVLT_ERR_ARG);
    BY_DTYPE(dtype,
             hipLaunchKernelGGL(ce_bwd_kernel<float>, dim3(rows), dim3(256), 0, STREAM(stream), (const float*)logits, (long)ld, rows, V, labels, lse, count, grad_scale, grad_scale_dev, (float*)dlogits, rows_dev),
             hipLaunchKernelGGL(ce_bwd_kernel<bf16_t>, dim3(rows), dim3(256), 0, STREAM(stream), (const bf16_t*)logits, (long)ld, rows, V, labels, lse, count, grad_scale, grad_scale_dev, (bf16_t*)dlogits, rows_dev));
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}

extern "C" int mvlt_softmax_rows(int dtype, const void* x, int64_t ld, int rows, int V, float* out, void* stream) {
    MVLT_CHECK(x && out && rows > 0 && V > 0 && ld >= V, MVLT_ERR_ARG);
    BY_DTYPE(dtype, hipLaunchKernelGGL(softmax_rows_kernel<float>, dim3(rows), dim3(64), 0, STREAM(stream), (const float*)x, (long)ld, V, out),
             hipLaunchKernelGGL(softmax_rows_kernel<bf16_t>, dim3(rows), dim3(64), 0, STREAM(stream), (const bf16_t*)x, (long)ld, V, out));
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}

extern "C" int mvlt_adamw(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                          int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                          float grad_scale, void* stream) {
    MVLT_CHECK(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, MVLT_ERR_ARG);
    MVLT_CHECK(aligned16(param) && aligned16(grad) && aligned16(exp_avg) && aligned16(exp_avg_sq), MVLT_ERR_ARG);
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    // TWO 256-thread workgroups per CU, not thousands: the sweep is seven interleaved streams (four read, three and a half written)
    // and HBM serves them best when few waves walk them -- stand-alone 6.0-6.1 TB/s with 256-768 workgroups against 5.3 with 8192
    // (scripts/adamw_probe.hip, 182.4 M parameters: 0.90 vs 1.03 ms); in the step 512 is the best of 256 / 384 / 512 / 1024 / 8192
    // (optimizer phase 0.845 vs 0.88 ms on a box with fast HBM, step -0.08 ms on a slower one; profiles/r5_adamw_grid.txt)
    static const int cap = [] { const char* e = getenv("MVLT_ADAMW_BLOCKS"); return e && atoi(e) > 0 ? atoi(e) : 512; }();
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n / 4 + 1, 256, cap)), dim3(256), 0, STREAM(stream), param, grad, exp_avg,
                       exp_avg_sq, (bf16_t*)shadow_bf16, (long)n, lr, beta1, beta2, eps, weight_decay, (float)bc1,
                       (float)sqrt(bc2), grad_scale);
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}

extern "C" int mvlt_attn_cached(const MvltAttnCached* p, void* stream) {
    MVLT_CHECK(p && p->qkv_new && p->k_cache && p->v_cache && p->out, MVLT_ERR_ARG);
    MVLT_CHECK(p->hd > 0 && p->hd <= 64 && p->B > 0 && p->nH > 0 && p->n_new > 0, MVLT_ERR_ARG);
    // with a device-side `past` the host cannot check the bound: the caller guarantees *past_dev + n_new <= cache_cap
    MVLT_CHECK(p->past_dev || p->past + p->n_new <= p->cache_cap, MVLT_ERR_ARG);
    const bool fast = p->hd == 64 && p->n_new <= CACHED_MAXNEW &&
                      aligned16(p->qkv_new) && aligned16(p->k_cache) && aligned16(p->v_cache) && aligned16(p->out);
    if (fast) {
        dim3 grid(p->B * p->nH);
        // bf16: a key block is 64 keys; four waves cover the <= 202 keys of a 150-token report in one round trip each
        if (p->cache_cap > 64 && p->dtype == MVLT_BF16)
            hipLaunchKernelGGL((attn_cached_kernel<bf16_t, 4>), grid, dim3(256), 0, STREAM(stream), *p);
        else
            BY_DTYPE(p->dtype, hipLaunchKernelGGL((attn_cached_kernel<float, 1>), grid, dim3(64), 0, STREAM(stream), *p),
                     hipLaunchKernelGGL((attn_cached_kernel<bf16_t, 1>), grid, dim3(64), 0, STREAM(stream), *p));
    } else {
        dim3 grid(p->B * p->nH * p->n_new);
        BY_DTYPE(p->dtype, hipLaunchKernelGGL(attn_cached_serial_kernel<float>, grid, dim3(64), 0, STREAM(stream), *p),
                 hipLaunchKernelGGL(attn_cached_serial_kernel<bf16_t>, grid, dim3(64), 0, STREAM(stream), *p));
    }
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}
extern "C" int mvlt_argmax(int dtype, const void* logits, int64_t ld, int rows, int V, int64_t* out, void* stream) {
    MVLT_CHECK(logits && out && rows > 0 && V > 0 && ld >= V, MVLT_ERR_ARG);
    BY_DTYPE(dtype, hipLaunchKernelGGL(argmax_kernel<float>, dim3(rows), dim3(256), 0, STREAM(stream), (const float*)logits, (long)ld, V, out),
             hipLaunchKernelGGL(argmax_kernel<bf16_t>, dim3(rows), dim3(256), 0, STREAM(stream), (const bf16_t*)logits, (long)ld, V, out));
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}

extern "C" int mvlt_image_normalize(const uint8_t* hwc, float* chw, int B, int H, int W, void* stream) {
    MVLT_CHECK(hwc && chw && B > 0 && H > 0 && W > 0, MVLT_ERR_ARG);
    MVLT_CHECK((long)H * W < (1L << 24), MVLT_ERR_ARG);            // the u64 sums of squares cannot overflow far below this
    hipLaunchKernelGGL(image_normalize_kernel, dim3(3, B), dim3(256), 0, STREAM(stream), hwc, chw, H, W);
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}
extern "C" int mvlt_mlm_mask(const MvltMlmMask* p, void* stream) {
    MVLT_CHECK(p && p->ids_in && p->full_len && p->ids_out && p->labels, MVLT_ERR_ARG);
    MVLT_CHECK(p->B > 0 && p->T > 0 && p->T <= 1024 && p->vocab_size > 0, MVLT_ERR_ARG);
    hipLaunchKernelGGL(mlm_mask_kernel, dim3(p->B), dim3(64), 0, STREAM(stream), *p);
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}

namespace {
struct PrefetchBatch { int n; MvltRange r[8]; };
__global__ __launch_bounds__(256) void prefetch_kernel(const PrefetchBatch b, unsigned* never) {
    unsigned acc = 0;
    for (int i = 0; i < b.n; ++i) {
        const char* base = reinterpret_cast<const char*>(b.r[i].ptr);
        const long lines = (b.r[i].bytes + 127) >> 7;
        for (long l = (long)blockIdx.x * 256 + threadIdx.x; l < lines; l += (long)gridDim.x * 256) {
            const long off = l << 7;
            acc ^= *reinterpret_cast<const unsigned*>(base + (off + 4 <= b.r[i].bytes ? off : (b.r[i].bytes - 4) & ~3L));
        }
    }
    if (acc == 0x9e3779b9u && never) *never = acc;          // keeps the loads alive; `never` is a null pointer
}
}  // namespace

extern "C" int mvlt_prefetch(const MvltRange* items, int n, void* stream) {
    MVLT_CHECK(items && n >= 1 && n <= 8, MVLT_ERR_ARG);
    PrefetchBatch b{};
    b.n = n;
    long lines = 0;
    for (int i = 0; i < n; ++i) { MVLT_CHECK(items[i].ptr && items[i].bytes >= 4, MVLT_ERR_ARG); b.r[i] = items[i]; lines += (items[i].bytes + 127) >> 7; }
    long blocks = (lines + 511) / 512;
    if (blocks > 256) blocks = 256;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(prefetch_kernel, dim3((unsigned)blocks), dim3(256), 0, STREAM(stream), b, (unsigned*)nullptr);
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}

extern "C" int mvlt_zero_batch(const MvltZeroItem* items, int n, void* stream) {
    MVLT_CHECK(items && n >= 0, MVLT_ERR_ARG);
    for (int i0 = 0; i0 < n; i0 += 32) {
        ZeroBatch b{};
        b.n = n - i0 < 32 ? n - i0 : 32;
        long mx = 1;
        for (int i = 0; i < b.n; ++i) { MVLT_CHECK(items[i0 + i].ptr && items[i0 + i].n >= 0, MVLT_ERR_ARG); b.it[i] = items[i0 + i]; if (b.it[i].n > mx) mx = b.it[i].n; }
        hipLaunchKernelGGL(zero_batch_kernel, dim3(grid_for(mx, 256, 64), b.n), dim3(256), 0, STREAM(stream), b);
        MVLT_LAUNCH_CHECK();
    }
    return MVLT_OK;
}

// Diagnostic (tests only): workgroups that occupy CUs -- each holds `lds_bytes` of LDS and spins on the 100 MHz real-time
// clock for `usec` microseconds.  Every wave reaches the exit condition by itself (a clock bound, no inter-block waiting).
namespace {
__global__ __launch_bounds__(256) void hold_cus_kernel(long long ticks, int lds_bytes, unsigned* never) {
    extern __shared__ __attribute__((aligned(16))) char hold_smem[];
    if (lds_bytes >= 4) reinterpret_cast<volatile int*>(hold_smem)[threadIdx.x % (lds_bytes / 4)] = (int)threadIdx.x;
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (lds_bytes >= 4 && reinterpret_cast<volatile int*>(hold_smem)[0] == 0x7fffffff && never) *never = 1;
}
}  // namespace

// Diagnostic (bench.py's one-GPU rehearsal of the multi-GPU step): a copy with the launch geometry of a ring collective -- FEW
// persistent workgroups, each streaming its slice with a bounded number of loads in flight -- so that it moves `bytes` at a
// few hundred GB/s for milliseconds beside the backward pass, the way RCCL's kernel does at world size 8 (xGMI-bound), instead
// of finishing at HBM speed.  dst may equal src (values unchanged).
namespace {
__global__ __launch_bounds__(256) void stream_copy_kernel(u32x4* dst, const u32x4* src, long n16, int inflight) {
    const long per = (n16 + gridDim.x - 1) / gridDim.x;
    const long lo = (long)blockIdx.x * per, hi = lo + per < n16 ? lo + per : n16;
    for (long i = lo + threadIdx.x; i < hi; i += 256L * 4) {
        u32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) if (j < inflight && i + 256L * j < hi) v[j] = __builtin_nontemporal_load(src + i + 256L * j);
#pragma unroll
        for (int j = 0; j < 4; ++j) if (j < inflight && i + 256L * j < hi) __builtin_nontemporal_store(v[j], dst + i + 256L * j);
        if (inflight < 4) for (int j = inflight; j < 4; ++j) if (i + 256L * j < hi) dst[i + 256L * j] = src[i + 256L * j];
    }
}
}  // namespace

extern "C" int mvlt_debug_stream_copy(void* dst, const void* src, int64_t bytes, int blocks, int inflight, void* stream) {
    MVLT_CHECK(dst && src && bytes > 0 && bytes % 16 == 0 && aligned16(dst) && aligned16(src), MVLT_ERR_ARG);
    MVLT_CHECK(blocks >= 1 && blocks <= 1024 && inflight >= 1 && inflight <= 4, MVLT_ERR_ARG);
    hipLaunchKernelGGL(stream_copy_kernel, dim3(blocks), dim3(256), 0, STREAM(stream), (u32x4*)dst, (const u32x4*)src, (long)(bytes / 16), inflight);
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}

extern "C" int mvlt_debug_hold_cus(int blocks, int lds_bytes, int usec, void* stream) {
    MVLT_CHECK(blocks >= 1 && blocks <= 4096 && lds_bytes >= 0 && lds_bytes <= 160 * 1024 && usec >= 0 && usec <= 5000000, MVLT_ERR_ARG);
    auto k = hold_cus_kernel;
    if (lds_bytes > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return MVLT_ERR_LAUNCH;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), (size_t)lds_bytes, STREAM(stream), (long long)usec * 100, lds_bytes, (unsigned*)nullptr);
    MVLT_LAUNCH_CHECK();
    return MVLT_OK;
}
