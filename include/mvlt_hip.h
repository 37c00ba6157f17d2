/* mvlt_hip.h -- C-ABI of the MI355X (gfx950) kernels for the MVLT hot path.
 *
 * The reference (Control-xl/Medical-Vision-Langauge-Transformer) has no
 * FFI/plugin seam: its hot path is eager PyTorch.  This
 * header is therefore the boundary a maintainer would bind (ctypes stub in
 * INTEGRATION.md): one entry point per fused op, each citing the reference
 * lines whose arithmetic it replaces.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless
 *    marked host; the caller owns every buffer (inputs, outputs, saved
 *    tensors, workspace) -- the library allocates nothing.
 *  - `stream` is a hipStream_t passed as void*; every launch goes on it.
 *  - return 0 on success, a negative MVLT_ERR_* otherwise; no exceptions.
 *  - re-entrant, no global mutable state; one process per GPU.
 *  - `dtype` selects storage/MFMA input type of activations and of the
 *    compute copies of weight matrices: MVLT_F32 (exact f32 MFMA,
 *    v_mfma_f32_16x16x4_f32) or MVLT_BF16 (v_mfma_f32_16x16x32_bf16, f32
 *    accumulate).  Biases, LayerNorm affine parameters, the relative position
 *    bias table, statistics and gradients of parameters are always f32.
 */
#ifndef MVLT_HIP_H
#define MVLT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { MVLT_F32 = 0, MVLT_BF16 = 1 };
enum { MVLT_OK = 0, MVLT_ERR_ARG = -1, MVLT_ERR_LAUNCH = -2, MVLT_ERR_UNSUPPORTED = -3 };

/* loader checks.  MVLT_ABI_VERSION is bumped on EVERY change of a struct layout or of an entry point's
 * signature; a binding compiles / hard-codes the value it was written against and compares it with what the
 * loaded library returns.  mvlt_sizeof(MVLT_STRUCT_*) lets a binding that mirrors the structs by hand (ctypes,
 * cgo, JNI) prove that its mirror has the size the library was compiled with (0 for an unknown id). */
#define MVLT_ABI_VERSION 8
int mvlt_version(void);            /* MVLT_ABI_VERSION of the loaded library */
const char* mvlt_arch(void);       /* "gfx950" */
enum { MVLT_STRUCT_GEMM = 0, MVLT_STRUCT_LAYERNORM = 1, MVLT_STRUCT_LAYERNORM_BWD = 2, MVLT_STRUCT_LN_REDUCE_ITEM = 3,
       MVLT_STRUCT_ATTN = 4, MVLT_STRUCT_SWIN_WMSA = 5, MVLT_STRUCT_EMBED = 6, MVLT_STRUCT_ATTN_CACHED = 7,
       MVLT_STRUCT_ZERO_ITEM = 8, MVLT_STRUCT_RANGE = 9, MVLT_STRUCT_MLM_MASK = 10, MVLT_STRUCT_GREEDY_STATE = 11, MVLT_STRUCT_SWIN_DBIAS_ITEM = 12, MVLT_STRUCT_COUNT = 13 };
size_t mvlt_sizeof(int struct_id);

/* ------------------------------------------------------------------ GEMM
 * C[M,N] = epilogue(A[M,K] * B[K,N]).  Replaces every nn.Linear on the path:
 * qkv/proj (visual_feature_extractor.py:231,252), Mlp fc1/fc2 (:135-141),
 * PatchMerging.reduction (:443), PatchEmbed.proj as im2col GEMM (:562), HF
 * BERT query/key/value/dense (modeling_bert.py:175-177,289,337,348), pooler,
 * MLM transform/decoder (:466-506) and all their dgrad/wgrad products.
 *   a_kmajor = 0: A[m*lda + k]          1: A[k*lda + m]   (wgrad: A = dY^T)
 *   b_kmajor = 0: B[n*ldb + k] (torch Linear weight [N,K])   1: B[k*ldb + n]
 * Epilogue, in this order (v = accumulator, m' = rowmap ? rowmap[m] : m):
 *   BIAS      v += bias[n]
 *   GELU      (SAVE_PRE: pre[m',n] = v)  v = gelu_erf(v)
 *   DROPOUT   v = keep(seed,tag,m*N+n) ? v/(1-p) : 0
 *   ROWSCALE  v *= rowscale[m' / rows_per_scale]            (DropPath)
 *   MUL_GELU_GRAD  v *= gelu'(aux[m',n])
 *   RESIDUAL  v += residual[m'*ldr + n]
 *   ACCUM     v += C[m',n]
 *   store C[m'*ldc + n]  (dtype, or f32 when OUT_F32)
 * split_k > 1 computes K-slices into `workspace` (f32 slabs) and a second
 * kernel reduces them and applies the epilogue (deterministic, no atomics). */
enum {
    MVLT_EPI_BIAS = 1, MVLT_EPI_GELU = 2, MVLT_EPI_SAVE_PRE = 4, MVLT_EPI_DROPOUT = 8,
    MVLT_EPI_ROWSCALE = 16, MVLT_EPI_RESIDUAL = 32, MVLT_EPI_ROWMAP = 64,
    MVLT_EPI_MUL_GELU_GRAD = 128, MVLT_EPI_OUT_F32 = 256, MVLT_EPI_ACCUM = 512
};
typedef struct MvltGemm {
    int dtype, M, N, K;
    const void* A; int64_t lda; int a_kmajor;
    const void* B; int64_t ldb; int b_kmajor;
    void* C; int64_t ldc;
    int epilogue;
    const float* bias;
    void* pre;                       /* ldc */
    const void* residual; int64_t ldr;
    const void* aux;                 /* ldc */
    const float* rowscale; int rows_per_scale;
    const int32_t* rowmap;
    float dropout_p; uint64_t seed; uint32_t tag;
    int split_k; void* workspace; size_t workspace_bytes;
    float* a_colsum;                 /* optional (a_kmajor only): a_colsum[m] = sum_k A[k*lda+m], i.e. the bias
                                        gradient colsum(dY) fused into the wgrad GEMM dW = dY^T X */
    void* event_after_main;          /* optional hipEvent_t recorded on `stream` right after the main GEMM kernel
                                        (before the split-K reduce): lets a benchmark time that kernel alone */
    const int32_t* m_dev;            /* optional, DEVICE int: the number of valid storage rows of A, read by the kernel
                                        (ragged batches planned on the GPU, mvlt_pack_plan: no host sync).  a_kmajor=0:
                                        rows of A / C beyond it are neither read nor written (M is the upper bound the
                                        launch is sized for); a_kmajor=1 (weight gradients): the reduction stops there
                                        (K is the upper bound) */
    const void* prefetch; int64_t prefetch_bytes;
                                     /* optional: a read-only byte range (the weight matrix of the NEXT nn.Linear of the
                                        layer) that this launch pulls towards the caches while it runs -- every thread
                                        reads and drops one dword of a few of its 128-byte lines.  The optimizer's sweep
                                        evicts all weights from the Infinity Cache once per step; a product whose weight
                                        panel is already on its way runs 10-20 % faster (profiles/r2_weight_prefetch.txt) */
} MvltGemm;
int mvlt_gemm(const MvltGemm* p, void* stream);
size_t mvlt_gemm_workspace_bytes(const MvltGemm* p);
/* introspection: tile (= kernel instantiation gemm_kernel<dtype,bm,bn,a_kmajor,b_kmajor>) and split-K chosen for p */
int mvlt_gemm_plan(const MvltGemm* p, int* bm, int* bn, int* split_k);
/* n (<= 8) independent products in ONE launch -- the weight gradients of one layer (dW_i = dY_i^T X_i, the
 * backward of the nn.Linear calls of one SwinTransformerBlock / BertLayer).  All items must have both operands
 * k-major, the same dtype and output widths that are all multiples of 128 or all multiples of 96; the tile
 * lists are concatenated (gemm_group_kernel<dtype,64,bn>), no split-K, no workspace.
 * MVLT_ERR_UNSUPPORTED if the items do not qualify (launch them one by one then). */
int mvlt_gemm_group(const MvltGemm* items, int n, void* stream);
/* Optional workspace of a grouped launch: put it into items[0].workspace / workspace_bytes.  With it, groups with fewer
 * output tiles than CUs (every weight-gradient group of the B = 32 step) are cut into k-slices that meet through f32
 * slabs inside the one launch (the last arriver of a tile sums the slices in slice order: deterministic, no atomics). */
size_t mvlt_gemm_group_workspace_bytes(const MvltGemm* items, int n);
/* Last-row MLM head fused with the greedy pick (model.py:896-900): out_idx[m] = argmax_n (A W^T + bias)[m, n]
 * (first index on ties), out_val[m] = that maximum (f32, may be NULL); the logits are never stored.  M <= 64,
 * both operands k-contiguous (MVLT_ERR_UNSUPPORTED otherwise); only MVLT_EPI_BIAS is honoured; C is unused.
 * part_val / part_idx: scratch, M * ceil(N / 16) elements each. */
int mvlt_gemm_argmax(const MvltGemm* p, float* part_val, int32_t* part_idx, int64_t* out_idx, float* out_val,
                     void* stream);
/* The same product with the per-token bookkeeping of greedy_search (model.py:896-913) in the finishing launch, all on the
 * device: next = argmax; samples that have finished emit pad_id; unfinished[m] &= (next != eos_id); ids[m, *col] = next;
 * scores[m, *col] = the maximum logit; new_ids[m, 0] = next (the first of the two ids the next cached step feeds);
 * alive[*col] is raised to 1 by every sample that is still unfinished (the caller zeroes `alive` when a decode starts);
 * *past += 1 (optional); *col += 1.  One launch (a workgroup per row; the last one to arrive -- `ticket`, an int32 the caller
 * zeroes once -- advances col / past) instead of the argmax finish + ten one-line kernels per replayed decode step.
 * has_eos = 0: no EOS handling (unfinished / alive unused). */
typedef struct MvltGreedyState {
    int64_t* unfinished; int64_t eos_id, pad_id; int has_eos;
    int64_t* col; int32_t* past;
    int64_t* ids; int64_t ld_ids; float* scores; int64_t ld_scores; int64_t* alive; int64_t* new_ids; int64_t ld_new;
    int32_t* ticket;
} MvltGreedyState;
int mvlt_gemm_argmax_greedy(const MvltGemm* p, float* part_val, int32_t* part_idx, const MvltGreedyState* g, void* stream);

/* Decode step (model.py:82-108: 2 new tokens per sample): skinny product with the reduction split over workgroups:
 * acc[s][M][N] (f32, k_splits slabs) = A[:, k-slice s] B[:, k-slice s]^T, M <= 64, both operands k-contiguous, no epilogue;
 * k_splits workgroups share every 16-column tile, each WRITES the slab of its slice (plain stores: no float atomics, nothing
 * to zero, bit-reproducible).  Pair it with mvlt_layernorm_acc_fwd(nsplit = k_splits), which adds the slabs in slice order
 * and applies bias + residual + LayerNorm (the BertSelfOutput / BertOutput tail, modeling_bert.py:282-293,340-351). */
int mvlt_gemm_skinny_accum(const MvltGemm* p, float* acc, int k_splits, void* stream);

/* column sums: out[n] = sum_m x[m*ld + n]  (bias gradients), f32 out.
 * workspace: f32 [mvlt_colsum_workspace_rows(M)][N]. */
int mvlt_colsum(int dtype, const void* x, int64_t ld, int M, int N, float* out, int accumulate,
                float* workspace, void* stream);
int mvlt_colsum_workspace_rows(int M);

/* ------------------------------------------------------------------ LayerNorm
 * y = LN(x)*gamma + beta over the last dim C (nn.LayerNorm at
 * visual_feature_extractor.py:308,314,422,553,653 eps 1e-5; HF BERT
 * LayerNorm eps 1e-12, modeling_bert.py:287,346,480).
 *  - merge_H/W != 0: input row r = (b,i,j) is the PatchMerging gather
 *    cat(x[2i,2j], x[2i+1,2j], x[2i,2j+1], x[2i+1,2j+1]) of x:[B,H*W,C/4]
 *    (visual_feature_extractor.py:435-440).
 *  - out_rowmap: y row = out_rowmap[r] (fuses roll(-s)+window_partition,
 *    :360-367, into the store).
 *  - gelu: y = gelu(LN(x)) (Swin final norm + Conv_layer nn.GELU, model.py:232-235);
 *    y_pre (optional) receives LN(x) for the backward pass.
 *  - mean/rstd (optional, f32 [rows]) are saved for the backward pass. */
typedef struct MvltLayerNorm {
    int dtype, rows, C; float eps;
    const void* x; const float* gamma; const float* beta;
    void* y; void* y_pre; float* mean; float* rstd;
    const int32_t* out_rowmap; int merge_H, merge_W; int gelu;
    const int32_t* rows_dev;         /* optional, DEVICE int: valid rows (<= rows); see MvltGemm.m_dev */
} MvltLayerNorm;
int mvlt_layernorm_fwd(const MvltLayerNorm* p, void* stream);

/* backward: dx (+= dres) ; dgamma/dbeta f32 [C].
 *  - dy_rowmap: dy row for logical row r is dy[dy_rowmap[r]].
 *  - gelu: dy is the gradient of gelu(LN(x)); y_pre must be given.
 *  - merge_H/W: dx is scattered back to x:[B,H*W,C/4].
 * workspace: f32 [2][mvlt_layernorm_bwd_workspace_rows()][C]. */
typedef struct MvltLayerNormBwd {
    int dtype, rows, C;
    const void* dy; const int32_t* dy_rowmap;
    const void* x; const float* mean; const float* rstd; const float* gamma;
    const void* y_pre; int gelu;
    const void* dres; void* dx;
    int merge_H, merge_W;
    float* dgamma; float* dbeta; int accumulate;
    float* workspace;
    /* optional second output, the gradient entering the residual BRANCH that produced x's summand:
     * dz[dz_rowmap ? dz_rowmap[r] : r] = dropout_mask(dx[r]; p, seed, tag, idx r*C+c) * dz_rowscale[r / rows_per_scale]
     * (Swin: window-order scatter + DropPath scale; BERT: hidden-dropout backward) */
    void* dz; const int32_t* dz_rowmap; const float* dz_rowscale; int dz_rows_per_scale;
    float dz_dropout_p; uint64_t seed; uint32_t tag;
    int defer_param_reduce;          /* 1: leave the partial rows in `workspace`; reduce them later, batched */
    const int32_t* rows_dev;         /* optional, DEVICE int: valid rows (<= rows); see MvltGemm.m_dev */
    /* optional: dy is the SUM of dy_parts (2 .. 4) tensors of the same shape, dy_part_stride elements apart, starting at `dy` (the
     * partial qkv-dgrad products of mvlt_swin_wmsa2_bwd): added in f32 while the rows are loaded, the sum rounded to bf16 once.
     * bf16, C = 4 * lanes * chunks widths (96 .. 1024), no gelu / merge; else MVLT_ERR_UNSUPPORTED.  0 / 1: plain dy. */
    int dy_parts; int64_t dy_part_stride;
} MvltLayerNormBwd;
int mvlt_layernorm_bwd(const MvltLayerNormBwd* p, void* stream);
int mvlt_layernorm_bwd_workspace_rows(void);
/* y[r,:] = LayerNorm(sum_s acc[s][r,:] + bias + residual[r,:]) * gamma + beta, rows <= a few hundred, C <= 2048; acc = the
 * nsplit slabs [nsplit][rows][C] (f32) of mvlt_gemm_skinny_accum (read only).  residual may be NULL. */
int mvlt_layernorm_acc_fwd(int dtype, const float* acc, int nsplit, const float* bias, const void* residual, const float* gamma,
                           const float* beta, float eps, int rows, int C, void* y, void* stream);
/* deferred parameter-gradient reduction: one launch per 96 LayerNorms instead of one per LayerNorm.
 * items is a HOST array; workspace = the buffer given to mvlt_layernorm_bwd, nparts = mvlt_layernorm_bwd_nparts(rows, C). */
typedef struct MvltLnReduceItem { const float* workspace; int nparts, C; float* dgamma; float* dbeta; } MvltLnReduceItem;
int mvlt_layernorm_bwd_nparts(int rows, int C);
int mvlt_layernorm_param_reduce_batch(const MvltLnReduceItem* items, int n, void* stream);

/* ------------------------------------------------------------------ attention
 * One (sequence, head) problem per workgroup; Q,K,V come from a fused
 * projection buffer qkv[rows, 3*nH*hd] laid out [3][nH][hd] per row.
 * mode MVLT_ATTN_SWIN  : WindowAttention core (visual_feature_extractor.py:234-251):
 *      softmax(q*scale @ k^T + table[relidx(q,k)][h] + shift_mask) @ v, N=49,
 *      relidx and the 0/-100 shift mask (:318-344) computed in-kernel from
 *      (win_res, shift); sequences are windows, nW per image.
 * mode MVLT_ATTN_BIDIR : HF eager attention (modeling_bert.py:111-136) with the
 *      MVLBert bidirectional key mask cat(1,image_mask,1,text>0) (model.py:125-128)
 *      rebuilt in-kernel from text_ids (int64 [B,T]) and optional image_mask (u8 [B,n_img]);
 *      masked keys get -10000 (model.py:182).
 * mode MVLT_ATTN_SEQ2SEQ : key allowed iff k<=q or k<=obj_end (model.py:118-123).
 * Dropout on the probabilities (attention_probs_dropout_prob) uses the counter RNG.
 * lse (f32 [nseq,nH,L]) is saved for the backward pass. */
enum { MVLT_ATTN_SWIN = 0, MVLT_ATTN_BIDIR = 1, MVLT_ATTN_SEQ2SEQ = 2 };
typedef struct MvltAttn {
    int dtype, mode;
    int nseq, L, nH, hd;
    const void* qkv; void* out;          /* out: [nseq*L, nH*hd] */
    float* lse;
    float scale;
    /* swin */
    const float* bias_table; int nW, win_res, shift;
    /* bert */
    const int64_t* text_ids; int T; const uint8_t* image_mask; int obj_end;
    float dropout_p; uint64_t seed; uint32_t tag;
    /* backward only */
    const void* dout; void* dqkv; float* dbias_table; float* delta_ws;
    /* packed rows (optional, MVLBert modes): sequence s occupies rows [row_start[s], row_start[s]+seq_len[s])
     * of qkv/out/dout/dqkv, seq_len[s] <= L; trailing zero-padded caption positions are simply absent
     * (they are masked keys in BIDIR mode and lie above the causal diagonal in SEQ2SEQ mode, so no kept
     * row ever reads them).  lse / delta_ws keep the [nseq,nH,L] layout.  NULL = dense [nseq*L] rows. */
    const int32_t* row_start; const int32_t* seq_len;
    /* backward only, MVLT_ATTN_SWIN only (optional): the output projection's dgrad inside the launch.  When non-NULL, `dout`
     * is the gradient of the PROJECTION's output ([nseq*L, nH*hd], window order) and dout_weight the projection weight
     * [nH*hd (out), nH*hd (in)] row-major in the compute dtype (WindowAttention.proj, visual_feature_extractor.py:252):
     * the kernel forms dO_h = dout . W[:, hd*h .. hd*h + hd - 1] per head itself (rounded to the compute dtype, as the
     * separate product would).  bf16, hd 32, nH 3 / 6 / 12, no attention dropout, shift 0 or 3; anything else:
     * MVLT_ERR_UNSUPPORTED. */
    const void* dout_weight;
    /* backward only (optional; honoured by the bf16 Swin launch, ignored elsewhere): a byte range a LATER kernel will stream,
     * same contract as MvltGemm.prefetch (one dword of every 128-byte line is read and dropped) */
    const void* prefetch; int64_t prefetch_bytes;
} MvltAttn;
int mvlt_attn_fwd(const MvltAttn* p, void* stream);
int mvlt_attn_bwd(const MvltAttn* p, void* stream);   /* delta_ws: f32 [nseq,nH,L] */
/* The same, and `event` (a hipEvent_t of the caller) completes with the call's LAST kernel: it rides on that dispatch as its
 * stop event instead of a marker packet behind it (an event record costs the recording stream ~5 us, this form ~2.5 us:
 * scripts/event_cost.hip).  For the fork of the weight-gradient stream, which follows the attention backward of every
 * BertLayer / Swin block (modeling_bert.py:282-293 backward; visual_feature_extractor.py:224-254 backward): the caller
 * then hipStreamWaitEvent()s on it.  Nothing is recorded when the call fails. */
int mvlt_attn_bwd_ev(const MvltAttn* p, void* stream, void* event);

/* ------------------------------------------------------------------ fused Swin (S)W-MSA (SURVEY.md 8b `swin_wmsa`)
 * The attention half of SwinTransformerBlock.forward in ONE launch (visual_feature_extractor.py:356-384 around
 * WindowAttention.forward :224-254):
 *   y = x + rowscale[b] * proj( softmax(q k^T * scale + rel_pos_bias + shift_mask) v ),  q,k,v = qkv(norm1(x))
 * with roll(-shift) + window_partition on the way in and window_reverse + roll(+shift) on the way out expressed
 * by the row map w2n (window-order row -> token row, the INT tables of SURVEY 8a2/8a3); window 7, head_dim 32.
 * x, y: [B*res*res, C] token order (y may not alias x).  The qkv tensor and the attention output never make an
 * HBM round trip.  Training: the optional pointers receive what the backward pass needs, as write-only outputs:
 *   xn_win   [B*res*res, C]  norm1(x), window order (A operand of the qkv weight gradient)
 *   attn_out [B*res*res, C]  attention output, window order (A operand of the proj weight gradient)
 *   lse      f32 [B*nW, nH, 49], mean / rstd f32 [B*res*res] (token order)
 * backward (mvlt_swin_wmsa_bwd): dy_win = gradient of the proj output in WINDOW order (rowscale already applied);
 *   output-projection dgrad, attention backward and qkv dgrad in one launch: reads q,k,v from qkv_win and lse (saved
 *   by the forward pass) and the transposed weight copies wproj_t / wqkv_t, writes dqkv [B*res*res, 3C] (window
 *   order, for the qkv weight gradient) and dxn_win [B*res*res, C] = dqkv Wqkv (window order; LayerNorm backward
 *   consumes it through its dy_rowmap); dbias_table f32 [169, nH] is ACCUMULATED (zero it first). */
typedef struct MvltSwinWmsa {
    int dtype, B, res, C, nH, shift;
    const void* x; void* y; const int32_t* w2n;
    const float* ln_gamma; const float* ln_beta; float ln_eps;
    const void* wqkv; const float* bqkv; const void* wproj; const float* bproj;
    const float* bias_table; float scale;
    const float* rowscale;             /* optional DropPath keep/(1-p), f32 [B] */
    void* xn_win; void* attn_out; float* lse; float* mean; float* rstd;
    /* backward only */
    const void* dy_win; void* dqkv; void* dxn_win; float* dbias_table;
    /* optional, training: q,k,v in the [B*res*res, 3C] window-order layout MvltAttn uses (forward: write-only
     * output, so the unfused mvlt_attn_bwd can run on a fused forward; backward: read instead of recomputed) */
    void* qkv_win;
    /* backward only: TRANSPOSED compute-dtype copies of the projection weights, wproj_t [C_in, C_out] = proj.weight^T and
     * wqkv_t [C, 3C] = qkv.weight^T (mvlt_transpose_batch makes them): the dgrad products then read k-contiguous rows */
    const void* wproj_t; const void* wqkv_t;
    /* forward only: 1 = one workgroup per (window, head group) and NO output projection: attn_out [B*res*res, C]
     * (window order, required) is the result, y / wproj / bproj / rowscale are unused; follow with mvlt_gemm
     * (bias, DropPath row scale, window-reverse row map, residual).  For launches with too few windows to fill the chip. */
    int head_split;
    /* mvlt_swin_wmsa2_bwd only: per-workgroup sums of the relative-position-bias gradient (see there), or NULL */
    float* dbias_ws;
} MvltSwinWmsa;
int mvlt_swin_wmsa_supported(int dtype, int C, int nH);   /* 1 when the fused kernels cover this width */
int mvlt_swin_wmsa_fwd(const MvltSwinWmsa* p, void* stream);
int mvlt_swin_wmsa_bwd(const MvltSwinWmsa* p, void* stream);
int mvlt_swin_wmsa_bwd_supported(int dtype, int C, int nH);

/* Second design of the fused forward (csrc/wmsa2.hip, bf16): a workgroup owns TWO windows (98 rows share every weight
 * fragment) and a GROUP of heads; the head groups of a window pair run in different workgroups and meet through
 * attn_out (write-through stores, one arrival flag per head group, sc1 loads) before each computes its own output columns of the
 * projection -- still ONE launch, and 256 workgroups at stage 2 of a B = 32 step where the first design has 128.
 * Same MvltSwinWmsa fields as mvlt_swin_wmsa_fwd (head_split ignored) with two differences: attn_out is REQUIRED (it is
 * the exchange buffer; eval callers pass scratch), and sync_ws is an int32 workspace of mvlt_swin_wmsa2_sync_words(B, res)
 * words that the caller zeroes ONCE when it allocates it: every launch leaves its counters zeroed.  Layout (independent
 * of B, so one workspace serves every launch of a stream): word 0 = STICKY ERROR COUNT, words 1..15 unused, then
 * {arrivals, readers done} per window pair.  One workspace per stream (launches that may overlap must not share counters).
 * Failure mode: the groups of a window pair wait for each other inside the launch, which needs them co-resident (the
 * launch is persistent with at most one workgroup per CU, grid <= #CUs and a multiple of the group count).  A wait is
 * bounded (2 s by default, mvlt_swin_wmsa2_set_timeout_ms); when it runs out the kernel adds 1 to word 0 AND writes NaN
 * into that unit's rows of y, so the failure reaches the loss even when nobody reads the word.  Callers read word 0 where
 * they synchronise anyway (PretrainStep does, one step late, without a sync) and raise.
 * Replaces visual_feature_extractor.py:224-254, 356-384 exactly like mvlt_swin_wmsa_fwd. */
int mvlt_swin_wmsa2_supported(int dtype, int B, int res, int C, int nH);
int mvlt_swin_wmsa2_sync_words(int B, int res);
int mvlt_swin_wmsa2_fwd(const MvltSwinWmsa* p, int32_t* sync_ws, void* stream);
/* The backward of the same block half in ONE launch, in the second design's shape (round 6): output-projection dgrad +
 * attention backward + qkv dgrad (visual_feature_extractor.py:224-254 backward); unit = (two windows, 3 heads), window order
 * throughout.  Reads dy_win [B*res*res, C] (gradient of the projection's output, rowscale applied), qkv_win, lse, wproj, wqkv,
 * bias_table, scale, shift; writes dqkv [B*res*res, 3C] and -- because the qkv dgrad sums over the heads and the head groups
 * are different workgroups that never meet -- nparts = mvlt_swin_wmsa2_bwd_parts() PARTIAL products
 *   dxn_win [nparts][B*res*res, C],  sum over the parts (f32) = dqkv Wqkv,
 * which mvlt_layernorm_bwd adds while it loads them (MvltLayerNormBwd.dy_parts); dbias_table f32 [169, nH] is ACCUMULATED.
 * bf16, C = 96 / 192 / 384 with nH = C / 32, shift 0 or 3, an even number of windows; else MVLT_ERR_UNSUPPORTED.
 * _ev: `event` completes with the kernel (as mvlt_attn_bwd_ev). */
int mvlt_swin_wmsa2_bwd_parts(int dtype, int B, int res, int C, int nH);    /* nH / 3, or 0 = shape not covered */
/* Relative-position-bias gradient of that launch: every workgroup stores the sums of its units' dS along the 169 diagonals for its
 * three heads into MvltSwinWmsa.dbias_ws, f32 [mvlt_swin_wmsa2_bwd_workgroups()][3 * 169] (plain stores: nothing to zero, no
 * atomics), and mvlt_swin_wmsa2_bwd_dbias ADDS the workgroups of up to any number of launches into their tables
 * (dbias_table f32 [169, nH]) in a fixed order -- one small launch for all Swin blocks of a step; items is a HOST array.
 * (MvltSwinWmsa.dbias_table itself is not touched by mvlt_swin_wmsa2_bwd; dbias_ws may be NULL: no bias gradient.) */
typedef struct MvltSwinDbiasItem { const float* ws; int nwg; int nH; float* dbias_table; } MvltSwinDbiasItem;
int mvlt_swin_wmsa2_bwd_workgroups(int dtype, int B, int res, int C, int nH);
int mvlt_swin_wmsa2_bwd_dbias(const MvltSwinDbiasItem* items, int n, void* stream);
int mvlt_swin_wmsa2_bwd(const MvltSwinWmsa* p, void* stream);
int mvlt_swin_wmsa2_bwd_ev(const MvltSwinWmsa* p, void* stream, void* event);
int mvlt_swin_wmsa2_set_timeout_ms(int ms);    /* bound of the hand-off wait, process-wide; 0 restores the default (2000) */
/* Diagnostic: `blocks` workgroups of 256 threads, each holding lds_bytes of LDS, spin for `usec` microseconds (100 MHz
 * real-time clock) and exit.  The tests use it to take CUs away from a launch that needs its workgroups co-resident. */
int mvlt_debug_hold_cus(int blocks, int lds_bytes, int usec, void* stream);
/* Diagnostic: copy `bytes` (multiple of 16, 16-byte aligned; dst may equal src) with `blocks` persistent workgroups of 256
 * threads and `inflight` (1..4) 16-byte loads in flight per thread -- the launch geometry of a ring collective's kernel,
 * i.e. a few hundred GB/s for milliseconds.  bench.py's one-GPU rehearsal of the data-parallel step issues it where the
 * all-reduce of a gradient bucket would run (MVLT_DDP_REHEARSE, mvlt_amd.ddp).  No reference counterpart (the reference
 * has no distributed code). */
int mvlt_debug_stream_copy(void* dst, const void* src, int64_t bytes, int blocks, int inflight, void* stream);

/* ------------------------------------------------------------------ data movement / embeddings
 * PatchEmbed im2col (visual_feature_extractor.py:562): img f32 NCHW [B,3,S,S]
 * -> cols [B*(S/P)^2, 3*P*P] in (c,dy,dx) order = Conv2d weight.view(96,-1). */
int mvlt_im2col_patch(int dtype, const float* img, void* cols, int B, int Cin, int S, int P, void* stream);

/* MVLBert.get_embedding (model.py:133-158): out[b,pos] = src(pos) + type_emb[pos<=obj_end] + pos_emb[pos]
 * with src = word_emb[cls] | image_feature | word_emb[sep] | word_emb[text];
 * NO LayerNorm/dropout.  Tables are f32 masters. */
typedef struct MvltEmbed {
    int dtype, B, n_img, T, H;
    const int64_t* text_ids;            /* [B,T] or NULL when T==0 */
    const void* image_feature;          /* [B,n_img,H] dtype */
    const float* word_emb; const float* pos_emb; const float* type_emb;
    int cls_id, sep_id, pos_offset, type_override;  /* cached step: pos_offset=past, type_override=0, n_img=-1 */
    void* out;                          /* [B, L, H] */
    /* backward */
    const void* dout; void* dimage; float* dword; float* dpos; float* dtype_emb;
    /* packed rows (optional): sequence b is written to / read from rows row_start[b] + pos, pos < seq_len[b] */
    const int32_t* row_start; const int32_t* seq_len;
    const int32_t* pos_offset_dev;      /* optional (forward): added to pos_offset, read on the device (replayed decode step) */
    /* backward: rows of the position / token-type tables (0 = unknown).  mvlt_embed_bwd OVERWRITES dpos rows
     * [pos_offset, pos_offset + L) and the dtype_emb rows in use with batch sums in a fixed order (no atomics,
     * bit-reproducible); with the row counts given it also zeroes every other row of the two tables, so the caller
     * clears only dword (which is accumulated with float atomics: a scatter-add over the batch's token ids). */
    int pos_rows, type_rows;
} MvltEmbed;
/* Packing plan of a ragged caption batch, computed ON THE DEVICE (no host sync): sample b keeps the sequence
 * positions [0, n_img + 2 + len_b) where len_b = 1 + the last caption position t with text_ids[b,t] != 0 or
 * (labels != NULL and labels[b,t] >= 0); the zero-padded tail beyond it is a masked key in the bidirectional mode
 * (model.py:125-128), lies above the causal diagonal in the seq2seq mode (:118-123) and carries no label, so no
 * kept row ever reads it.  Outputs: seq_len[b] = n_img + 2 + len_b, row_start[b] = exclusive prefix sum,
 * total_rows[0] = sum (feeds MvltGemm.m_dev / MvltLayerNorm.rows_dev), row_start64 = the same as int64 (index
 * tensors), text_row[b*T + t] = int64 packed row of caption position t (row_start[b] -- the [CLS] row, finite
 * data -- for positions beyond len_b: their label is -100 by construction).  B <= 65536. */
int mvlt_pack_plan(const int64_t* text_ids, const int64_t* labels, int B, int T, int n_img,
                   int32_t* row_start, int32_t* seq_len, int32_t* total_rows, int64_t* row_start64, int64_t* text_row,
                   void* stream);
/* MLM head on the labelled rows (F.cross_entropy(ignore_index=-100), model.py:410, only reads them): stable partition
 * of the N caption positions, labelled ones first.  gather_row[slot] = text_row[i] (the packed row of position i;
 * i itself when text_row is NULL), sel_labels[slot] = labels[i], count[0] = number of labelled positions (feeds
 * MvltGemm.m_dev).  One launch, no host sync; N <= 2^20. */
int mvlt_label_plan(const int64_t* labels, const int64_t* text_row, int N, int32_t* gather_row, int64_t* sel_labels,
                    int32_t* count, void* stream);
/* out[rowmap[i], :] = in[i, :] for i < min(rows, *count): backward of a row gather with unique source rows */
int mvlt_rows_scatter(int dtype, const void* in, void* out, int rows, int C, const int32_t* rowmap,
                      const int32_t* count, void* stream);
int mvlt_embed_fwd(const MvltEmbed* p, void* stream);
/* backward: ONLY dword is accumulated (float atomics over the batch's token ids, the [CLS] / [SEP] rows included): zero it
 * first.  dpos / dtype_emb are OVERWRITTEN with ordered batch sums -- a second call does not add to the first -- and rows
 * outside the range in use are zeroed only when pos_rows / type_rows are given (see the struct). */
int mvlt_embed_bwd(const MvltEmbed* p, void* stream);

/* out[i,:] = scale[i_src / rows_per_scale] * mask(in[src,:]) where src = rowmap ? rowmap[i] : i;
 * dropout mask (p>0) indexed by src*C + c.  Used for DropPath / dropout backward
 * and window-order gathers of gradients. */
int mvlt_rows_transform(int dtype, const void* in, void* out, int rows, int C, const int32_t* rowmap,
                        const float* rowscale, int rows_per_scale,
                        float dropout_p, uint64_t seed, uint32_t tag, void* stream);

/* elementwise helpers */
int mvlt_cast(int src_dtype, const void* src, int dst_dtype, void* dst, int64_t n, void* stream);
int mvlt_gelu_fwd(int dtype, const void* x, void* y, int64_t n, void* stream);
int mvlt_gelu_bwd(int dtype, const void* x, const void* dy, void* dx, int64_t n, void* stream); /* dx = dy*gelu'(x) */
int mvlt_softmax_rows(int dtype, const void* x, int64_t ld, int rows, int V, float* out, void* stream); /* f32 [rows,V] */
int mvlt_tanh_fwd(int dtype, const void* x, void* y, int64_t n, void* stream);
int mvlt_tanh_bwd(int dtype, const void* y, const void* dy, void* dx, int64_t n, void* stream);
int mvlt_dropout_mask(uint8_t* keep, int64_t n, float p, uint64_t seed, uint32_t tag, void* stream);
int mvlt_droppath_scale(float* scale, int B, float p, uint64_t seed, uint32_t tag, void* stream);
/* every DropPath scale of a forward pass in one launch (visual_feature_extractor.py:30-44 for each of the 2 x 24 residual
 * branches): scale [rows, B] f32, row r keeps sample b with probability 1 - probs[r] (device f32 [rows], each < 1) and
 * holds 1 / (1 - probs[r]) or 0; row r draws with tag + r */
int mvlt_droppath_scales(float* scale, const float* probs, int rows, int B, uint64_t seed, uint32_t tag, void* stream);

/* ------------------------------------------------------------------ losses
 * F.cross_entropy(logits, labels, ignore_index=-100) (model.py:410,418):
 * logits [rows, ld>=V] dtype; loss_sum/count are f32 scalars (device),
 * lse f32 [rows].  bwd writes dlogits = scale*(softmax - onehot) for valid
 * rows and 0 for ignored rows, where scale = grad_scale / max(count,1)
 * (mean reduction), in place over logits when dlogits == logits. */
int mvlt_ce_fwd(int dtype, const void* logits, int64_t ld, int rows, int V, const int64_t* labels,
                float* lse, float* loss_sum, float* count, void* stream);
int mvlt_ce_bwd(int dtype, const void* logits, int64_t ld, int rows, int V, const int64_t* labels,
                const float* lse, const float* count, float grad_scale, const float* grad_scale_dev,
                void* dlogits, void* stream);   /* grad_scale_dev (optional): upstream dL/dloss on the device */
/* the same with the number of valid rows on the DEVICE (rows = upper bound; see MvltGemm.m_dev): rows beyond
 * *rows_dev are neither read nor written */
int mvlt_ce_fwd_ragged(int dtype, const void* logits, int64_t ld, int rows, int V, const int64_t* labels,
                       float* lse, float* loss_sum, float* count, const int32_t* rows_dev, void* stream);
int mvlt_ce_bwd_ragged(int dtype, const void* logits, int64_t ld, int rows, int V, const int64_t* labels,
                       const float* lse, const float* count, float grad_scale, const float* grad_scale_dev,
                       void* dlogits, const int32_t* rows_dev, void* stream);

/* ------------------------------------------------------------------ optimizer
 * torch.optim.AdamW step (run_pretrain.py:165-166: lr 4e-5, betas (0.9,0.999),
 * eps 1e-6, weight_decay 1e-4) over one flat f32 segment; also refreshes the
 * bf16 compute copy (shadow, may be NULL).  grad_scale multiplies g first
 * (1/world_size for DDP averaging). */
int mvlt_adamw(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
               int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
               int step, float grad_scale, void* stream);

/* ------------------------------------------------------------------ decode (greedy, KV cache)
 * 2-token cached step attention (model.py:82-108): q rows = n_new new tokens,
 * keys = cache[0..past) + new; causal over the new tokens. */
typedef struct MvltAttnCached {
    int dtype, B, nH, hd, past, n_new, cache_cap;
    const void* qkv_new;                /* [B*n_new, 3*nH*hd] */
    void* k_cache; void* v_cache;       /* [B, nH, cache_cap, hd]; new K/V appended at `past` */
    void* out;                          /* [B*n_new, nH*hd] */
    float scale;
    const int32_t* past_dev;            /* optional: `past` read from device memory instead (a decode loop that
                                           is replayed without host involvement keeps its position on the GPU) */
} MvltAttnCached;
int mvlt_attn_cached(const MvltAttnCached* p, void* stream);
/* argmax over V of logits [rows, ld] -> int64 ids (greedy_search, model.py:896-900) */
int mvlt_argmax(int dtype, const void* logits, int64_t ld, int rows, int V, int64_t* out, void* stream);

/* zero up to 32 small f32 buffers in ONE launch (the relative-position-bias gradients of all Swin blocks are
 * accumulated with atomics and must start from zero; items is a HOST array) */
typedef struct MvltZeroItem { float* ptr; int64_t n; } MvltZeroItem;
int mvlt_zero_batch(const MvltZeroItem* items, int n, void* stream);

/* Pull up to 8 read-only byte ranges (the weight matrices of the layer about to run) towards the GPU's caches: one
 * dword of every 128-byte line is read and dropped.  The optimizer's sweep over 6 GB of state evicts every weight from
 * the 256 MB Infinity Cache once per step, so each nn.Linear of the reference (model.py / visual_feature_extractor.py
 * forward passes) would otherwise stream its weight panels from HBM at one miss latency per k-tile.  items is a HOST
 * array; nothing is written. */
typedef struct MvltRange { const void* ptr; int64_t bytes; } MvltRange;
int mvlt_prefetch(const MvltRange* items, int n, void* stream);

/* ------------------------------------------------------------------ input pipeline
 * The reference prepares every sample on the host (run_pretrain_rgc_roco_medicat.py:94-212); these two entry
 * points move the per-step arithmetic of that code to the GPU for pre-resized / pre-tokenised shards.
 *
 * Image normalisation (:107-110, :127-130): HWC uint8 RGB [B,H,W,3] -> CHW f32 [B,3,H,W] with, per image and
 * channel, (x - mean_c) / var_c -- the reference divides by np.var (population VARIANCE, not the standard
 * deviation); a constant channel gives 0/0 = NaN exactly as numpy does.  Sums are exact integers. */
int mvlt_image_normalize(const uint8_t* hwc, float* chw, int B, int H, int W, void* stream);
/* MLM masking, _random_mask_word (:188-212) + the truncation of :170-176, on already truncated id rows:
 * n = min(10, max(1, round(0.2 * full_len[b]))) distinct token positions drawn uniformly from [0, full_len[b]);
 * each: 80 % -> mask_id, 10 % -> uniform random id in [0, vocab_size), 10 % unchanged; label = original id.
 * Position i of the untruncated caption lives at column i (i < T-1 or full_len <= T), the last token ([END]) at
 * column T-1 when full_len > T; other positions were cut off and their masks are dropped, like the reference.
 * Rows with itm_label[b] == 0 are left unmasked (:160-164).  Counter RNG (seed, row, position): statistically
 * equivalent to the reference's Python `random`, not bit-identical.  ids_in/ids_out/labels: int64 [B,T]. */
typedef struct MvltMlmMask {
    int B, T, vocab_size, mask_id;
    const int64_t* ids_in; const int32_t* full_len; const int64_t* itm_label;   /* itm_label may be NULL */
    int64_t* ids_out; int64_t* labels;
    uint64_t seed;
} MvltMlmMask;
int mvlt_mlm_mask(const MvltMlmMask* p, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MVLT_HIP_H */
