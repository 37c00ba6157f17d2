"""Greedy report generation with a KV cache, drop-in for the reference's UniLM
decode loop (``modules/model.py:82-108`` cache branch of ``get_embedding``,
``:577-604`` prepare_inputs, ``:826-984`` greedy_search, ``:890-894`` cache trim).

Layout: one preallocated cache ``[layers][B, nH, cap, hd]`` per K and V
(cap = n_img + 2 + max_length + 1).  Step 0 runs the full seq2seq forward over
``[CLS] img [SEP] [MASK]``; every later step feeds the 2 tokens
``[last_token, [MASK]]`` at positions ``past, past+1`` (type 0), appends their
K/V in place and attends causally (``mvlt_attn_cached``).  "Trimming the [MASK]
slot" (model.py:890-894) is just ``past += 1``: the next step overwrites it.
Greedy mode replays one captured HIP graph per token (``_GreedyGraph``: position,
output column and finished flags live on the device; ``MVLT_DECODE_GRAPH=0`` selects
the eager loop, which 'sample' mode always uses).
Beam search (``beam_search``): same cached steps over B*beams rows, cache rows gathered by beam
index, scorer bookkeeping restated from HF transformers 4.16 (parity with the reference unpinned).
"""
from __future__ import annotations

import os

import torch

from . import _lib as L
from . import ops
from .arena import Arena
from .bert import EncoderOutput
from .runtime import compute_dtype_of


# the two N = H products of a decode layer split their reduction over workgroups; the k-slices meet in f32 slabs that the
# LayerNorm launch behind them adds in slice order (round 5: no float atomics -- bf16 greedy decoding is reproducible run to run)
_SKINNY_SPLIT = True
_SPLITS = (2, 4)      # reduction splits of (attention output, FFN-out) projections: the fastest of the round-2 sweep


def _layers_cached(mv, ar, x, kc, vc, past, n_new, out_last=None):
    """x: [B*n_new, H] embeddings of the new tokens -> last hidden [B*n_new, H] (written into ``out_last`` when given)."""
    cfg = mv.config
    H = cfg.hidden_size
    nH = cfg.num_attention_heads
    rows = x.shape[0]
    # the two N = H products of a layer (attention output, FFN-out) split their reduction over workgroups and meet in
    # an f32 accumulator; bias + residual + LayerNorm read it (and zero it again) in the launch that follows anyway
    # (bf16 only: the exact-f32 parity mode keeps the deterministic summation order of the plain skinny kernel)
    split = _SKINNY_SPLIT and rows <= 64 and x.dtype == torch.bfloat16
    acc = None
    if split:
        key = ("decode_acc", rows, H, x.device.index)
        acc = ar._views.get(key)
        if acc is None:
            acc = ar._views[key] = torch.empty((max(_SPLITS), rows, H), dtype=torch.float32, device=x.device)
    nl = len(mv.encoder.layer)
    for i, layer in enumerate(mv.encoder.layer):
        last_out = out_last if i == nl - 1 else None
        sa, so = layer.attention.self, layer.attention.output
        qkv = ops.gemm(x, ar.compute(sa.query.weight, 3 * H), bias=ar.master_span(sa.query.bias, 3 * H))
        ctx = ops.attn_cached(qkv, kc[i], vc[i], past, (H // nH) ** -0.5)
        if split:
            ops.gemm_skinny_accum(ctx, ar.compute(so.dense.weight), acc[:_SPLITS[0]], _SPLITS[0])
            x1 = ops.layernorm_acc_fwd(acc[:_SPLITS[0]], so.dense.bias.data, x, so.LayerNorm.weight.data, so.LayerNorm.bias.data,
                                       so.LayerNorm.eps, x.dtype)
        else:
            y1 = ops.gemm(ctx, ar.compute(so.dense.weight), bias=so.dense.bias.data, residual=x)
            x1, _, _, _ = ops.layernorm_fwd(y1, so.LayerNorm.weight.data, so.LayerNorm.bias.data, so.LayerNorm.eps,
                                            save_stats=False)
        a = ops.gemm(x1, ar.compute(layer.intermediate.dense.weight), bias=layer.intermediate.dense.bias.data, gelu=True)
        lo = layer.output
        if split:
            ops.gemm_skinny_accum(a, ar.compute(lo.dense.weight), acc[:_SPLITS[1]], _SPLITS[1])
            x = ops.layernorm_acc_fwd(acc[:_SPLITS[1]], lo.dense.bias.data, x1, lo.LayerNorm.weight.data, lo.LayerNorm.bias.data,
                                      lo.LayerNorm.eps, x1.dtype, out=last_out)
        else:
            y2 = ops.gemm(a, ar.compute(lo.dense.weight), bias=lo.dense.bias.data, residual=x1)
            x, _, _, _ = ops.layernorm_fwd(y2, lo.LayerNorm.weight.data, lo.LayerNorm.bias.data, lo.LayerNorm.eps,
                                           save_stats=False, out=last_out)
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


def _no_graph():
    return None


class _GreedyGraph:
    """The whole per-token work of greedy decoding -- last-row MLM head + argmax + [END]/PAD bookkeeping + the
    2-token cached forward -- captured ONCE as a HIP graph and replayed per token.  Everything a replay needs
    lives in static device buffers: the cache position (``past``: read by the embedding and attention kernels
    through ``pos_offset_dev`` / ``past_dev``), the output column index, the ids fed to the next step, the
    unfinished flags, and the [B, max_length] output matrices.  The host only replays and, every 8 tokens, reads
    the all-finished flags back."""

    def __reduce__(self):                 # captured graphs do not survive pickling: rebuilt on the next call
        return (_no_graph, ())

    def __init__(self, model, B, n_img, max_length, cd, pad, eos, mask_id, key):
        mv, cfg = model.MVLBert, model.config
        dev = next(model.parameters()).device
        H, nH = cfg.hidden_size, cfg.num_attention_heads
        nl = len(mv.encoder.layer)
        self.key, self.model, self.B, self.max_length, self.eos, self.pad = key, model, B, max_length, eos, pad
        cap = n_img + 2 + max_length + 1
        self.kc = [torch.zeros((B, nH, cap, H // nH), dtype=cd, device=dev) for _ in range(nl)]
        self.vc = [torch.zeros((B, nH, cap, H // nH), dtype=cd, device=dev) for _ in range(nl)]
        self.past = torch.zeros(1, dtype=torch.int32, device=dev)
        self.col = torch.zeros(1, dtype=torch.int64, device=dev)
        self.new_ids = torch.full((B, 2), mask_id, dtype=torch.int64, device=dev)       # [last token, MASK]
        # last hidden states of the two new tokens of every sample; the MLM head reads the [MASK] rows in place (row stride 2 H)
        self.hfull = torch.zeros((B * 2, H), dtype=cd, device=dev)
        self.hlast = self.hfull.view(B, 2, H)[:, 1]
        self.unfinished = torch.ones(B, dtype=torch.int64, device=dev)
        self.ids = torch.zeros((B, max_length), dtype=torch.int64, device=dev)
        self.scores = torch.zeros((B, max_length), dtype=torch.float32, device=dev)
        self.alive = torch.ones(max_length, dtype=torch.int64, device=dev)
        self.cd, self.graph = cd, None
        # device-side state of the greedy loop, handed to mvlt_gemm_argmax_greedy: the pick, PAD for finished samples, the
        # EOS flags, the ids / scores columns, the next input id, `past` and `col` are all advanced by its finishing launch
        st = self.state = L.MvltGreedyState()
        st.unfinished, st.eos_id, st.pad_id, st.has_eos = self.unfinished.data_ptr(), (eos if eos is not None else -1), pad, int(eos is not None)
        st.col, st.past = self.col.data_ptr(), self.past.data_ptr()
        st.ids, st.ld_ids, st.scores, st.ld_scores = self.ids.data_ptr(), max_length, self.scores.data_ptr(), max_length
        st.alive, st.new_ids, st.ld_new = self.alive.data_ptr(), self.new_ids.data_ptr(), 2
        self.ticket = torch.zeros(1, dtype=torch.int32, device=dev)
        st.ticket = self.ticket.data_ptr()

    def head(self):
        """token <- argmax(MLM head(hlast)); record it in column `col`; past += 1; col += 1 (all on the device)."""
        model = self.model
        ar = Arena.of(model, self.cd)
        hd = model.MLM_head_seq2seq
        _, _, t2, _, _ = hd._transform(ar, self.hlast, False)
        # decoder GEMM fused with the greedy pick and its bookkeeping: the [B, 30522] logits are never written
        ops.gemm_argmax_greedy(t2, ar.compute(hd.predictions.decoder.weight), hd.predictions.decoder.bias.data, self.state)

    def forward2(self):
        """2-token cached forward of [last token, MASK] at positions past, past+1 (model.py:82-108).  `past` was advanced by
        the pick that produced the token: the previous step's [MASK] slot is overwritten (model.py:890-894)."""
        mv = self.model.MVLBert
        ar = Arena.of(self.model, self.cd)
        B, H = self.B, mv.config.hidden_size
        x = _embed_new(mv, self.new_ids, self.past, self.cd).view(B * 2, H)
        _layers_cached(mv, ar, x, self.kc, self.vc, self.past, 2, out_last=self.hfull)

    def capture(self):
        # scratch buffers (split-K workspace) used inside the graph get their own tag: the captured pointers must
        # never be freed or handed to other work by a later, larger request on the main stream
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), ops.on_stream(side, "graph"):     # warm-up outside the capture (allocator, lazy state)
            self.head(); self.forward2()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            with ops.on_stream(torch.cuda.current_stream(), "graph"):
                self.head(); self.forward2()
        self.graph = g


def _greedy_graph_loop(model, feat, max_length, pad, eos, mask_id, cd):
    mv, cfg = model.MVLBert, model.config
    B, n_img, H = feat.shape
    nH = cfg.num_attention_heads
    nl = len(mv.encoder.layer)
    ar = Arena.of(model, cd)
    key = (B, n_img, max_length, cd, pad, eos, mask_id, ar.flat.data_ptr(), feat.device.index)
    gg = model.__dict__.get("_mvlt_greedy_graph")
    if gg is None or gg.key != key:
        gg = _GreedyGraph(model, B, n_img, max_length, cd, pad, eos, mask_id, key)
        gg.capture()
        model.__dict__["_mvlt_greedy_graph"] = gg
    # ---- step 0: [CLS] img [SEP] [MASK], full seq2seq forward (model.py:110-160), eager
    mask_col = gg.new_ids[:, 1:2].contiguous()
    hidden, _, saved = mv._forward(feat, mask_col, mask_col, None, True, True)
    L0 = n_img + 3
    for i in range(nl):
        _fill_cache_from_qkv(saved["layers"][i][1], B, L0, nH, H // nH, gg.kc[i], gg.vc[i], L0 - 1)
    del saved
    gg.past.fill_(L0 - 2); gg.col.zero_(); gg.unfinished.fill_(1); gg.alive.zero_(); gg.ticket.zero_()      # (the first pick advances `past` to L0 - 1; the picks raise `alive`)
    gg.hlast.copy_(hidden[:, -1])
    done = 0
    for t in range(max_length - 1):
        gg.graph.replay()                        # token t, then the forward that prepares token t+1
        done = t + 1
        if eos is not None and done % 8 == 0 and 0 in gg.alive[done - 8:done].tolist():   # one host sync per 8 tokens
            break
    else:
        gg.head()                                # last token: head only
        done = max_length
    flags = gg.alive[:done].tolist() if eos is not None else []
    if 0 in flags:       # cut where the reference's per-token check stops; it breaks before appending that step's score
        n_out = flags.index(0) + 1
        n_scores = n_out - 1
    else:
        n_out = n_scores = done
    ids = gg.ids[:, :n_out].clone()
    scores = gg.scores[:, :n_scores].t().reshape(-1).clone() if n_scores > 0 else torch.empty(0, device=feat.device)
    return ids, scores


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
    # (the fused decoder-GEMM + argmax of the graph path holds the whole batch in one 64-row tile)
    if sample_mode == 'greedy' and os.environ.get("MVLT_DECODE_GRAPH", "1") == "1" and feat.shape[0] <= 64:
        return _greedy_graph_loop(model, feat, max_length, pad, eos, mask_id, cd)
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
        if sample_mode == 'greedy' and t2.shape[0] <= 64:
            return ops.gemm_argmax(t2, ar.compute(head.predictions.decoder.weight), head.predictions.decoder.bias.data)
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


# ----------------------------------------------------------------------------- beam search (model.py:636-816)
class BeamHypotheses:
    """n-best list of finished hypotheses of one sample (HF transformers 4.16 ``BeamHypotheses``)."""

    def __init__(self, num_beams, length_penalty, early_stopping):
        self.num_beams, self.length_penalty, self.early_stopping = num_beams, length_penalty, early_stopping
        self.beams = []
        self.worst_score = 1e9

    def __len__(self):
        return len(self.beams)

    def add(self, hyp, sum_logprobs):
        score = sum_logprobs / (len(hyp) ** self.length_penalty)
        if len(self) < self.num_beams or score > self.worst_score:
            self.beams.append((score, hyp))
            if len(self) > self.num_beams:
                ranked = sorted((s, i) for i, (s, _) in enumerate(self.beams))
                del self.beams[ranked[0][1]]
                self.worst_score = ranked[1][0]
            else:
                self.worst_score = min(score, self.worst_score)

    def is_done(self, best_sum_logprobs, cur_len):
        if len(self) < self.num_beams:
            return False
        if self.early_stopping:
            return True
        return self.worst_score >= best_sum_logprobs / cur_len ** self.length_penalty


class BeamScorer:
    """The bookkeeping of HF ``BeamSearchScorer`` as of transformers 4.16 (``process`` / ``finalize``) with the
    reference's construction arguments (model.py:505-507: length_penalty 1.0, early stopping off, one hypothesis
    kept per sample).  The reference pins transformers only as ``>=4.16.0`` and the installed 5.x no longer has
    the class, so this is restated from the 4.16 semantics: parity with the reference is UNPINNED.
    Works on host lists; the device tensors are read back once per step, as the HF scorer does (`.item()`)."""

    def __init__(self, batch_size, num_beams, length_penalty=1.0, do_early_stopping=False):
        self.num_beams = num_beams
        self.hyps = [BeamHypotheses(num_beams, length_penalty, do_early_stopping) for _ in range(batch_size)]
        self.done = [False] * batch_size

    @property
    def is_done(self):
        return all(self.done)

    def process(self, input_ids, next_scores, next_tokens, next_indices, pad_token_id, eos_token_id):
        """input_ids: list[B*beams] of token lists; next_*: [B][2*beams] lists.  Returns three flat lists."""
        nb = self.num_beams
        cur_len = len(input_ids[0])
        out_s, out_t, out_i = [], [], []
        for b, hyp in enumerate(self.hyps):
            if self.done[b]:
                out_s += [0.0] * nb; out_t += [pad_token_id] * nb; out_i += [0] * nb
                continue
            kept = 0
            for rank, (tok, sc, idx) in enumerate(zip(next_tokens[b], next_scores[b], next_indices[b])):
                row = b * nb + idx
                if eos_token_id is not None and tok == eos_token_id:
                    if rank >= nb:
                        continue
                    hyp.add(list(input_ids[row]), sc)
                else:
                    out_s.append(sc); out_t.append(tok); out_i.append(row)
                    kept += 1
                if kept == nb:
                    break
            if kept < nb:
                raise ValueError(f"At most {nb} tokens in {next_tokens[b]} can be equal to `eos_token_id`")
            self.done[b] = self.done[b] or hyp.is_done(max(next_scores[b]), cur_len)
        return out_s, out_t, out_i

    def finalize(self, input_ids, final_beam_scores, max_length, pad_token_id, eos_token_id):
        nb = self.num_beams
        for b, hyp in enumerate(self.hyps):
            if self.done[b]:
                continue
            for k in range(nb):
                hyp.add(list(input_ids[b * nb + k]), final_beam_scores[b * nb + k])
        best = [sorted(h.beams, key=lambda x: x[0])[-1][1] for h in self.hyps]
        lengths = [len(h) for h in best]
        width = min(max(lengths) + 1, max_length)
        out = [[pad_token_id] * width for _ in best]
        for i, h in enumerate(best):
            out[i][:len(h)] = h[:width]
            if len(h) < max_length and len(h) < width:
                out[i][len(h)] = eos_token_id
        return out


@torch.no_grad()
def beam_search(model, image_feature, num_beams, learning_strategy='unilm', max_length=None, pad_token_id=None,
                eos_token_id=None):
    """``MVLBertForImageCaption.beam_search`` (model.py:636-816) with the KV cache: scores = log_softmax(logits) +
    beam score, top 2*beams over (beam, token), scorer bookkeeping, cache rows gathered by ``beam_idx`` (:758-763).
    Step 0 runs once per image (all beams of a sample start identical and only beam 0 carries score 0, :681-682),
    then 2-token cached steps over the B*beams rows.  Returns the sequences [B, <= max_length] (:795-815)."""
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
    nb = num_beams
    cap = n_img + 2 + max_length + 1
    scorer = BeamScorer(B, nb)

    def logp_of(hlast):
        _, _, t2, _, _ = head._transform(ar, hlast.contiguous(), False)
        logits, _ = head._logits(ar, t2)
        return torch.log_softmax(logits[:, :V].float(), dim=-1)

    # ---- step 0 on the B images; caches replicated to the beams afterwards
    mask_col = torch.full((B, 1), mask_id, dtype=torch.int64, device=dev)
    hidden, _, saved = mv._forward(feat, mask_col, mask_col, None, True, True)
    L0 = n_img + 3
    past = L0 - 1
    rep = torch.arange(B, device=dev).repeat_interleave(nb)
    kc, vc = [], []
    for i in range(nl):
        k1 = torch.zeros((B, nH, cap, hd), dtype=cd, device=dev)
        v1 = torch.zeros((B, nH, cap, hd), dtype=cd, device=dev)
        _fill_cache_from_qkv(saved["layers"][i][1], B, L0, nH, hd, k1, v1, past)
        kc.append(k1.index_select(0, rep)); vc.append(v1.index_select(0, rep))
    del saved
    logp = logp_of(hidden[:, -1]).index_select(0, rep)                        # [B*nb, V]
    beam_scores = torch.zeros((B, nb), dtype=torch.float32, device=dev)
    beam_scores[:, 1:] = -1e9
    beam_scores = beam_scores.view(-1)
    input_ids = [[mask_id] for _ in range(B * nb)]        # what the reference hands the scorer at step 0 (:701-702)
    mask2 = torch.full((B * nb, 1), mask_id, dtype=torch.int64, device=dev)
    cur_len = 0
    while cur_len < max_length:
        scores = (logp + beam_scores[:, None]).view(B, nb * V)
        top_s, top_t = torch.topk(scores, 2 * nb, dim=1, largest=True, sorted=True)
        top_i = torch.div(top_t, V, rounding_mode="floor")
        top_t = top_t % V
        s_l, t_l, i_l = scorer.process(input_ids, top_s.tolist(), top_t.tolist(), top_i.tolist(), pad, eos)
        beam_scores = torch.tensor(s_l, dtype=torch.float32, device=dev)
        beam_tok = torch.tensor(t_l, dtype=torch.int64, device=dev)
        beam_idx = torch.tensor(i_l, dtype=torch.int64, device=dev)
        input_ids = [[t] for t in t_l] if cur_len == 0 else [input_ids[i] + [t] for i, t in zip(i_l, t_l)]
        cur_len += 1
        if scorer.is_done or cur_len >= max_length:
            break
        for i in range(nl):                                  # beam reorder of the cache (model.py:758-763)
            kc[i] = kc[i].index_select(0, beam_idx)
            vc[i] = vc[i].index_select(0, beam_idx)
        new_ids = torch.cat([beam_tok[:, None], mask2], dim=1)
        x = _embed_new(mv, new_ids, past, cd).view(B * nb * 2, H)
        h = _layers_cached(mv, ar, x, kc, vc, past, 2).view(B * nb, 2, H)
        past += 1
        logp = logp_of(h[:, -1])
    seqs = scorer.finalize(input_ids, beam_scores.tolist(), cfg.max_length, pad, eos)
    return torch.tensor(seqs, dtype=torch.int64, device=dev)
