"""Greedy report generation with a KV cache, drop-in for the reference's UniLM
decode loop (``modules/model.py:82-108`` cache branch of ``get_embedding``,
``:577-604`` prepare_inputs, ``:826-984`` greedy_search, ``:890-894`` cache trim).

Layout: one preallocated cache ``[layers][B, nH, cap, hd]`` per K and V
(cap = n_img + 2 + max_length + 1).  Step 0 runs the full seq2seq forward over
``[CLS] img [SEP] [MASK]``; every later step feeds the 2 tokens
``[last_token, [MASK]]`` at positions ``past, past+1`` (type 0), appends their
K/V in place and attends causally (``mvlt_attn_cached``).  "Trimming the [MASK]
slot" (model.py:890-894) is just ``past += 1``: the next step overwrites it.
Beam search is not built.
"""
from __future__ import annotations

import torch

from . import ops
from .arena import Arena
from .bert import EncoderOutput
from .runtime import compute_dtype_of


def _layers_cached(mv, ar, x, kc, vc, past, n_new):
    """x: [B*n_new, H] embeddings of the new tokens -> last hidden [B*n_new, H]."""
    cfg = mv.config
    H = cfg.hidden_size
    nH = cfg.num_attention_heads
    for i, layer in enumerate(mv.encoder.layer):
        sa, so = layer.attention.self, layer.attention.output
        qkv = ops.gemm(x, ar.compute(sa.query.weight, 3 * H), bias=ar.master_span(sa.query.bias, 3 * H))
        ctx = ops.attn_cached(qkv, kc[i], vc[i], past, (H // nH) ** -0.5)
        y1 = ops.gemm(ctx, ar.compute(so.dense.weight), bias=so.dense.bias.data, residual=x)
        x1, _, _, _ = ops.layernorm_fwd(y1, so.LayerNorm.weight.data, so.LayerNorm.bias.data, so.LayerNorm.eps,
                                        save_stats=False)
        a = ops.gemm(x1, ar.compute(layer.intermediate.dense.weight), bias=layer.intermediate.dense.bias.data, gelu=True)
        y2 = ops.gemm(a, ar.compute(layer.output.dense.weight), bias=layer.output.dense.bias.data, residual=x1)
        x, _, _, _ = ops.layernorm_fwd(y2, layer.output.LayerNorm.weight.data, layer.output.LayerNorm.bias.data,
                                       layer.output.LayerNorm.eps, save_stats=False)
    return x


def _embed_new(mv, ids, past, dtype):
    cfg = mv.config
    return ops.embed_fwd(ids.contiguous(), None, mv.word_embeddings.weight.data, mv.position_embeddings.weight.data,
                         mv.token_type_embeddings.weight.data, cfg.cls_token_id, cfg.sep_token_id, dtype=dtype,
                         pos_offset=past, type_override=0)


def _fill_cache_from_qkv(qkv, B, Lq, nH, hd, kc, vc, keep):
    v5 = qkv.view(B, Lq, 3, nH, hd)
    kc[:, :, :keep].copy_(v5[:, :keep, 1].permute(0, 2, 1, 3))
    vc[:, :, :keep].copy_(v5[:, :keep, 2].permute(0, 2, 1, 3))


@torch.no_grad()
def cached_forward(mv, text_idx, image_feature, past_key_values, seq2seq_mask):
    """``MVLBert.forward(..., past_key_values=..., use_cache=True)`` API (model.py:59-62):
    returns ``(EncoderOutput(last_hidden_state, past_key_values), pooler_output)``."""
    cd = compute_dtype_of(mv)
    ar = Arena.of(mv, cd)
    ar.refresh_shadow()
    cfg = mv.config
    H, nH = cfg.hidden_size, cfg.num_attention_heads
    hd = H // nH
    nl = len(mv.encoder.layer)
    if past_key_values is None:
        feat = image_feature.to(cd).contiguous()
        B, n_img, _ = feat.shape
        ids = text_idx.contiguous() if text_idx is not None else None
        hidden, pooled, saved = mv._forward(feat, ids, ids, None, bool(seq2seq_mask), True)
        Lq = hidden.shape[1]
        pkv = []
        for i in range(nl):
            v5 = saved["layers"][i][1].view(B, Lq, 3, nH, hd)
            pkv.append((v5[:, :, 1].permute(0, 2, 1, 3).contiguous(), v5[:, :, 2].permute(0, 2, 1, 3).contiguous()))
        return EncoderOutput(hidden, tuple(pkv)), pooled
    if not seq2seq_mask:
        raise NotImplementedError("the reference only uses the cache with seq2seq_mask=True (model.py:604)")
    B, n_new = text_idx.shape
    past = past_key_values[0][0].shape[2]
    cap = past + n_new
    kc = [torch.empty((B, nH, cap, hd), dtype=cd, device=text_idx.device) for _ in range(nl)]
    vc = [torch.empty((B, nH, cap, hd), dtype=cd, device=text_idx.device) for _ in range(nl)]
    for i in range(nl):
        kc[i][:, :, :past].copy_(past_key_values[i][0])
        vc[i][:, :, :past].copy_(past_key_values[i][1])
    x = _embed_new(mv, text_idx, past, cd).view(B * n_new, H)
    h = _layers_cached(mv, ar, x, kc, vc, past, n_new).view(B, n_new, H)
    return EncoderOutput(h, tuple((kc[i], vc[i]) for i in range(nl))), None


@torch.no_grad()
def greedy_search(model, image_feature, learning_strategy='unilm', sample_mode='greedy', max_length=None,
                  pad_token_id=None, eos_token_id=None):
    """Returns ``(input_ids [B, n_steps], token_scores)`` like the reference
    (model.py:984: scores of all but the final step, concatenated along dim -1)."""
    if learning_strategy != 'unilm':
        raise NotImplementedError("only learning_strategy='unilm' is coherent with the KV cache (SURVEY.md 3.3)")
    mv, cfg = model.MVLBert, model.config
    cd = compute_dtype_of(model)
    ar = Arena.of(model, cd)
    ar.refresh_shadow()
    max_length = max_length if max_length is not None else cfg.max_length
    pad = pad_token_id if pad_token_id is not None else cfg.pad_token_id
    eos = eos_token_id if eos_token_id is not None else cfg.eos_token_id
    tok = getattr(model, "tokenizer", None)
    mask_id = tok.mask_token_id if tok is not None else cfg.mask_token_id
    feat = image_feature.to(cd).contiguous()
    B, n_img, H = feat.shape
    nH = cfg.num_attention_heads
    hd = H // nH
    nl = len(mv.encoder.layer)
    dev = feat.device
    head = model.MLM_head_seq2seq
    V = head.predictions.decoder.out_features
    cap = n_img + 2 + max_length + 1
    kc = [torch.zeros((B, nH, cap, hd), dtype=cd, device=dev) for _ in range(nl)]
    vc = [torch.zeros((B, nH, cap, hd), dtype=cd, device=dev) for _ in range(nl)]
    mask_col = torch.full((B, 1), mask_id, dtype=torch.int64, device=dev)

    def next_from(hlast):
        pre, t1, t2, _, _ = head._transform(ar, hlast.contiguous(), False)
        logits, _ = head._logits(ar, t2)
        if sample_mode == 'greedy':
            nxt = ops.argmax(logits, V)
            score = logits[:, :V].float().gather(1, nxt[:, None]).squeeze(1)
        elif sample_mode == 'sample':
            probs = ops.softmax_rows(logits, V)
            nxt = torch.multinomial(probs, num_samples=1, replacement=True).squeeze(1)
            score = torch.log(probs.gather(1, nxt[:, None])).squeeze(1)
        else:
            raise ValueError("sample mode error!")
        return nxt, score

    # ---- step 0: [CLS] img [SEP] [MASK], full seq2seq forward (model.py:110-160)
    hidden, _, saved = mv._forward(feat, mask_col, mask_col, None, True, True)
    L0 = n_img + 3
    past = L0 - 1
    for i in range(nl):
        _fill_cache_from_qkv(saved["layers"][i][1], B, L0, nH, hd, kc[i], vc[i], past)
    del saved
    unfinished = torch.ones(B, dtype=torch.int64, device=dev)
    ids_cols, scores, alive = [], [], []
    hlast = hidden[:, -1]
    cur_len = 0
    n_out = None                     # number of generated columns once every sequence has emitted [END]
    # The reference asks the device "all finished?" after every token (model.py:954), which serialises host and
    # GPU.  Finished sequences only emit PAD, so running a few steps past the end changes nothing that is kept:
    # the flag of every step is recorded on the device, read back every `sync_every` steps, and the outputs are
    # cut at the exact step the reference would have stopped at.
    sync_every = 8
    while cur_len < max_length:
        nxt, score = next_from(hlast)
        if eos is not None:
            nxt = nxt * unfinished + pad * (1 - unfinished)
            unfinished = unfinished * (nxt != eos).long()
            alive.append(unfinished.max())
        ids_cols.append(nxt[:, None])
        if eos is not None and (len(alive) % sync_every == 0 or cur_len + 1 >= max_length):
            flags = torch.stack(alive[-sync_every:]).tolist()          # one host sync per `sync_every` tokens
            if 0 in flags:
                n_out = len(alive) - len(flags) + flags.index(0) + 1
                break
        cur_len += 1
        scores.append(score)
        if cur_len >= max_length:
            break
        new_ids = torch.cat([nxt[:, None], mask_col], dim=1)              # [last_token, MASK] (model.py:587-591)
        x = _embed_new(mv, new_ids, past, cd).view(B * 2, H)
        h = _layers_cached(mv, ar, x, kc, vc, past, 2).view(B, 2, H)
        past += 1                                                         # drop the [MASK] slot (model.py:890-894)
        hlast = h[:, -1]
    if n_out is not None:            # the reference breaks before appending the score of the finishing step
        ids_cols, scores = ids_cols[:n_out], scores[:n_out - 1]
    input_ids = torch.cat(ids_cols, dim=-1) if ids_cols else None
    token_scores = torch.cat(scores, dim=-1) if scores else torch.empty(0, device=dev)
    return input_ids, token_scores
