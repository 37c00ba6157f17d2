"""MVLBert single-stream encoder, drop-in for reference ``modules/model.py:16-183``
plus the HF ``BertEncoder`` / ``BertPooler`` blocks it imports (model.py:5;
arithmetic per transformers modeling_bert.py:111-136,164-203,282-293,325-351,451-463).

The nn.Module tree holds parameters under the reference/HF state-dict names;
the arithmetic is a kernel sequence, 7 launches per layer forward:

    QKV GEMM (one [3H,H] GEMM: query/key/value weights are adjacent in the arena)
    fused attention (bidirectional key mask from text ids / seq2seq mask from
        (row, col, obj_end), -10000 additive, probability dropout in-kernel)
    out-proj GEMM (+bias, hidden dropout, residual) | LayerNorm(1e-12)
    FFN-in GEMM (+bias, GELU, pre-activation saved)
    FFN-out GEMM (+bias, hidden dropout, residual) | LayerNorm(1e-12)
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .arena import Arena
from .runtime import backward_begin, compute_dtype_of, next_seed


# ----------------------------------------------------------------------------- parameter holders (HF names)
class BertSelfAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        H = config.hidden_size
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = H // config.num_attention_heads
        self.query, self.key, self.value = nn.Linear(H, H), nn.Linear(H, H), nn.Linear(H, H)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)

    def _arena_groups(self):
        return [[self.query.weight, self.key.weight, self.value.weight],
                [self.query.bias, self.key.bias, self.value.bias]]


class BertSelfOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)


class BertAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self = BertSelfAttention(config)
        self.output = BertSelfOutput(config)


class BertIntermediate(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)


class BertOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)


class BertLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.attention = BertAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)


class BertEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.layer = nn.ModuleList([BertLayer(config) for _ in range(config.num_hidden_layers)])


class BertPooler(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.activation = nn.Tanh()


class EncoderOutput(tuple):
    """Stand-in for HF BaseModelOutputWithPastAndCrossAttentions: supports
    ``out[0]``, ``.last_hidden_state`` and ``.past_key_values`` (model.py:62,:694,:759)."""

    def __new__(cls, last_hidden_state, past_key_values=None):
        o = super().__new__(cls, (last_hidden_state, past_key_values))
        o.last_hidden_state = last_hidden_state
        o.past_key_values = past_key_values
        return o


class _EncFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, token, image_feature, mod, text_idx, mask_ids, image_mask, seq2seq, save, pack=None):
        with ops.pin_stream():
            hidden, pooled, saved = mod._forward(image_feature, text_idx, mask_ids, image_mask, seq2seq, save, pack)
        ctx.mod, ctx.saved = mod, saved
        ctx.set_materialize_grads(False)     # unused pooled output -> None, so the pooler gets no gradient
        if pooled is None:
            pooled = hidden.new_zeros(1)
            ctx.mark_non_differentiable(pooled)
        return hidden, pooled

    @staticmethod
    def backward(ctx, dhidden, dpooled):
        with ops.pin_stream():
            dimg = ctx.mod._backward(ctx.saved, dhidden, dpooled)
        ctx.saved = None
        return None, dimg, None, None, None, None, None, None, None


class MVLBert(nn.Module):
    """Signature of reference ``MVLBert`` (model.py:16-72)."""

    def __init__(self, config, add_pooling_layer=False):
        super().__init__()
        self.config = config
        self.word_embeddings = nn.Embedding(config.vocab_size + 1, config.hidden_size)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.embedding_LayerNorm = nn.LayerNorm(config.hidden_size, eps=1e-12)   # constructed, never applied (model.py:25,:158)
        self.embedding_dropout = nn.Dropout(config.hidden_dropout_prob)
        self.encoder = BertEncoder(config)
        self.is_decoder = getattr(config, "is_decoder", False)
        self.pooler = BertPooler(config) if add_pooling_layer else None
        self.register_buffer("position_ids", torch.arange(512).expand((1, -1)))
        if config.hidden_size // config.num_attention_heads != 64:
            raise NotImplementedError("attention kernel is built for head_dim 64 (bert-base)")
        self.last_seed = None

    # ------------------------------------------------------------------ public forward (model.py:35-72)
    def forward(self, text_idx, text_mask, image_feature, image_mask, past_key_values=None, use_cache=False,
                seq2seq_mask=False, output_text_image_seperate=False):
        if not image_feature.is_cuda:
            raise RuntimeError("mvlt_amd runs on the GPU only (no CPU fallback)")
        if past_key_values is not None or use_cache:
            from .decode import cached_forward
            return cached_forward(self, text_idx, image_feature, past_key_values, seq2seq_mask)
        cd = compute_dtype_of(self)
        B, n_img, _ = image_feature.shape
        T = 0 if text_idx is None else text_idx.shape[1]
        obj_end = n_img + 1
        text_end = obj_end + T + 1
        tok = self.__dict__.get("_mvlt_token")
        if tok is None or tok.device != image_feature.device:
            tok = torch.zeros(1, device=image_feature.device, requires_grad=True)
            self.__dict__["_mvlt_token"] = tok
        mask_ids = None
        if text_idx is not None:
            text_idx = text_idx.contiguous()
            # the bidirectional key mask is "text_mask"; callers pass (text_idx > 0) (model.py:337,:384)
            mask_ids = text_idx if text_mask is None else text_mask.to(torch.int64).contiguous()
        im = None
        if image_mask is not None and image_mask.dtype != torch.bool:
            im = (image_mask != 0)
        elif image_mask is not None:
            im = image_mask
        im_u8 = im.to(torch.uint8).contiguous() if im is not None else None
        feat = image_feature if image_feature.dtype == cd else image_feature.to(cd)
        hidden, pooled = _EncFn.apply(tok, feat.contiguous(), self, text_idx, mask_ids, im_u8, bool(seq2seq_mask),
                                      torch.is_grad_enabled())
        pooler_output = pooled if self.pooler is not None else None
        if output_text_image_seperate:
            return (hidden[:, obj_end + 1:text_end], hidden[:, 1:obj_end], pooler_output, hidden[:, obj_end])
        return EncoderOutput(hidden), pooler_output

    # ------------------------------------------------------------------ packed rows (pre-training fast path)
    def forward_packed(self, text_idx, image_feature, text_lengths, seq2seq_mask=False):
        """Same encoder on PACKED rows: sample b keeps positions [0, n_img + 2 + text_lengths[b]) and its
        trailing zero-padded caption positions are not materialised at all.  Those positions are masked keys
        in the bidirectional mode (model.py:125-128) and lie above the causal diagonal in the seq2seq mode
        (model.py:118-123), so every kept row is computed from exactly the same operands as in the dense
        layout; the dropped rows are the ones the reference computes and never reads (padding carries no
        label).  ``text_lengths``: host int tensor/list [B] (tokeniser output; no device sync is needed).
        Returns (hidden [R, H], pooled [B, H], row_start int64 [B], seq_len int32 [B]) -- the last two on the
        device; seq_len counts [CLS] + image tokens + [SEP] + kept caption positions."""
        cd = compute_dtype_of(self)
        B, n_img, _ = image_feature.shape
        T = text_idx.shape[1]
        dev = image_feature.device
        lens = torch.as_tensor(text_lengths, dtype=torch.int32).reshape(-1).clamp(0, T) + (n_img + 2)
        if lens.numel() != B:
            raise ValueError("text_lengths must have one entry per sample")
        starts = torch.cumsum(lens, 0, dtype=torch.int32) - lens
        R = int(lens.sum())
        both = torch.stack((starts, lens)).pin_memory().to(dev, non_blocking=True)      # no sync: pinned, async
        pack = (both[0], both[1], R, both[0].to(torch.int64))
        tok = self.__dict__.get("_mvlt_token")
        if tok is None or tok.device != dev:
            tok = torch.zeros(1, device=dev, requires_grad=True)
            self.__dict__["_mvlt_token"] = tok
        text_idx = text_idx.contiguous()
        feat = image_feature if image_feature.dtype == cd else image_feature.to(cd)
        hidden, pooled = _EncFn.apply(tok, feat.contiguous(), self, text_idx, text_idx, None, bool(seq2seq_mask),
                                      torch.is_grad_enabled(), pack)
        return hidden, (pooled if self.pooler is not None else None), pack[3], pack[1]

    def forward_autopack(self, text_idx, image_feature, labels=None, seq2seq_mask=False, inputs_ready=None,
                         want_label_plan=False):
        """forward_packed with the plan computed ON THE DEVICE from the ids themselves (mvlt_pack_plan): no extra
        argument, no host sync.  Sample b keeps [CLS] img [SEP] and its caption up to the last position that holds a
        non-zero id (or a label); the launch geometry is sized for the dense upper bound B * L and every kernel
        reads the real row count from device memory (MvltGemm.m_dev / MvltLayerNorm.rows_dev).
        Returns (hidden [B*L, H] -- rows beyond the packed total are never written --, pooled [B, H],
        text_row int64 [B*T]: the packed row of every caption position, the sample's [CLS] row for dropped ones)."""
        cd = compute_dtype_of(self)
        B, n_img, _ = image_feature.shape
        dev = image_feature.device
        ids_in = text_idx
        text_idx = text_idx.contiguous()
        lab = None if labels is None else labels.reshape(text_idx.shape).to(torch.int64).contiguous()
        self.__dict__.pop("_mvlt_label_plan", None)
        same_inputs = (inputs_ready is not None and text_idx.data_ptr() == ids_in.data_ptr()
                       and (labels is None or lab.data_ptr() == labels.data_ptr()))
        if same_inputs:
            # ``inputs_ready``: an event recorded before the image tower was queued.  The ids / labels are the caller's own
            # tensors (no conversion kernel of ours produced them), so the two single-workgroup plan kernels (19 + 13 us)
            # run on the side stream behind that event, beside the tower, and the encoder waits for them
            side = ops.side_stream(dev)
            side.wait_event(inputs_ready)
            # (allocated under the side stream too: a block of the main stream's pool may still have a queued reader)
            with torch.cuda.stream(side), ops.on_stream(side, "side"):
                rs, sl, tot, rs64, trow = ops.pack_plan(text_idx, lab, n_img)
                if want_label_plan and lab is not None:
                    self.__dict__["_mvlt_label_plan"] = ops.label_plan(lab.reshape(-1), trow)
            done = torch.cuda.Event()
            done.record(side)
            torch.cuda.current_stream().wait_event(done)
            # the plan tensors live in the SIDE stream's allocator pool and are read by main-stream kernels: tell the
            # caching allocator, so a later side-stream allocation cannot reuse a block a queued main-stream kernel reads
            main = torch.cuda.current_stream()
            for t in (rs, sl, tot, rs64, trow) + tuple(self.__dict__.get("_mvlt_label_plan") or ()):
                if isinstance(t, torch.Tensor):
                    t.record_stream(main)
        else:
            with ops.pin_stream():
                rs, sl, tot, rs64, trow = ops.pack_plan(text_idx, lab, n_img)
        pack = (rs, sl, B * (n_img + 2 + text_idx.shape[1]), rs64, tot)
        tok = self.__dict__.get("_mvlt_token")
        if tok is None or tok.device != dev:
            tok = torch.zeros(1, device=dev, requires_grad=True)
            self.__dict__["_mvlt_token"] = tok
        feat = image_feature if image_feature.dtype == cd else image_feature.to(cd)
        hidden, pooled = _EncFn.apply(tok, feat.contiguous(), self, text_idx, text_idx, None, bool(seq2seq_mask),
                                      torch.is_grad_enabled(), pack)
        return hidden, (pooled if self.pooler is not None else None), trow

    # ------------------------------------------------------------------ engine
    def _forward(self, feat, text_idx, mask_ids, image_mask, seq2seq, save, pack=None):
        cfg = self.config
        cd = feat.dtype
        ar = Arena.of(self, cd)
        ar.refresh_shadow(tail=False)
        # a deferred optimizer tail (optim.FusedAdamW(defer_tail=True): BertLayers 1.., pooler, heads) starts HERE, on the
        # optimizer stream, behind the image tower and beside the encoder; a layer waits for its chunk right before it runs
        tail = ar.__dict__.get("_opt_tail")
        tail_at = None
        if tail is not None:
            tail.launch()
            tail_at = [ar.offset[id(layer.attention.self.query.weight)] + 1 for layer in self.encoder.layer]
        B, n_img, H = feat.shape
        T = 0 if text_idx is None else text_idx.shape[1]
        Lq = n_img + 2 + T
        nH = cfg.num_attention_heads
        train = self.training
        p_h = cfg.hidden_dropout_prob if train else 0.0
        p_a = cfg.attention_probs_dropout_prob if train else 0.0
        seed = next_seed() if (p_h > 0 or p_a > 0) else 0
        self.last_seed = seed
        # pack = (row_start, seq_len, R): activations are [R, H] with the trailing zero-padded caption positions
        # of every sample left out (forward_packed); otherwise dense [B*Lq, H]
        rows = B * Lq if pack is None else pack[2]
        rd = pack[4] if (pack is not None and len(pack) > 4) else None     # row count on the device (auto-packed batch)
        x = ops.embed_fwd(text_idx, feat, self.word_embeddings.weight.data, self.position_embeddings.weight.data,
                          self.token_type_embeddings.weight.data, cfg.cls_token_id, cfg.sep_token_id,
                          pack=pack).view(rows, H)
        mode = L.ATTN_SEQ2SEQ if seq2seq else L.ATTN_BIDIR
        akw = dict(text_ids=mask_ids, image_mask=image_mask, obj_end=n_img + 1, pack=pack)
        layers = []
        native = ops.NATIVE
        if native:
            # one native call per BertLayer (csrc/host.cpp bert_layer_fwd): same launches, ~4 us of host time each
            hx, st = ops.host(), ops.stream_int()
            attn_desc = self._attn_desc(mode, B, Lq, nH, mask_ids, T, image_mask, n_img + 1, pack)
            sd = ops.s64(seed)
            for i, layer in enumerate(self.encoder.layer):
                if tail_at is not None and ar.__dict__.get("_opt_tail") is not None:
                    tail.wait_for(tail_at[i])
                w, f, _ = self._layer_desc(ar, layer)
                out = hx.bert_layer_fwd(x, w, f, H, cfg.intermediate_size, layer.output.LayerNorm.eps, attn_desc,
                                        p_h, p_a, sd, i, save, st)
                x = out[0]
                if save:
                    layers.append(out[1:])
        for i, layer in enumerate(() if native else self.encoder.layer):
            if tail_at is not None and ar.__dict__.get("_opt_tail") is not None:
                tail.wait_for(tail_at[i])
            sa, so = layer.attention.self, layer.attention.output
            qkv = ops.gemm(x, ar.compute(sa.query.weight, 3 * H), bias=ar.master_span(sa.query.bias, 3 * H), m_dev=rd)
            ctx, lse = ops.attn_fwd(qkv, mode, B, Lq, nH, H // nH, (H // nH) ** -0.5,
                                    dropout=(p_a, seed, 8 * i + 0), **akw)
            y1 = ops.gemm(ctx, ar.compute(so.dense.weight), bias=so.dense.bias.data, dropout=(p_h, seed, 8 * i + 1),
                          residual=x, m_dev=rd)
            x1, m1, r1, _ = ops.layernorm_fwd(y1, so.LayerNorm.weight.data, so.LayerNorm.bias.data, so.LayerNorm.eps,
                                              save_stats=save, rows_dev=rd)
            h = torch.empty((rows, cfg.intermediate_size), dtype=cd, device=x.device)
            a = ops.gemm(x1, ar.compute(layer.intermediate.dense.weight), bias=layer.intermediate.dense.bias.data,
                         gelu=True, save_pre=h, m_dev=rd)
            y2 = ops.gemm(a, ar.compute(layer.output.dense.weight), bias=layer.output.dense.bias.data,
                          dropout=(p_h, seed, 8 * i + 2), residual=x1, m_dev=rd)
            x2, m2, r2, _ = ops.layernorm_fwd(y2, layer.output.LayerNorm.weight.data, layer.output.LayerNorm.bias.data,
                                              layer.output.LayerNorm.eps, save_stats=save, rows_dev=rd)
            if save:
                layers.append((x, qkv, ctx, lse, y1, m1, r1, x1, h, a, y2, m2, r2))
            x = x2
        hidden = x.view(B, Lq, H) if pack is None else x
        pooled = cls = None
        if tail is not None and ar.__dict__.get("_opt_tail") is not None:
            tail.wait_for(None)              # pooler and heads: the last chunk
        if self.pooler is not None:          # tanh(Linear(h[:,0]))  (modeling_bert.py:451-463)
            cls = hidden[:, 0] if pack is None else x.index_select(0, pack[3])
            pooled = ops.tanh_fwd(ops.gemm(cls, ar.compute(self.pooler.dense.weight),
                                           bias=self.pooler.dense.bias.data))
        saved = None
        if save:
            saved = dict(ar=ar, layers=layers, B=B, Lq=Lq, n_img=n_img, text_idx=text_idx, akw=akw, mode=mode,
                         seed=seed, p_h=p_h, p_a=p_a, cls=cls, pooled=pooled, pack=pack, rows=rows, native=native, T=T)
        return hidden, pooled, saved

    # ---- descriptors of the native host path: raw device pointers, built once per arena
    @staticmethod
    def _attn_desc(mode, B, Lq, nH, mask_ids, T, image_mask, obj_end, pack):
        ptr = lambda t: 0 if t is None else t.data_ptr()
        return [mode, B, Lq, nH, ptr(mask_ids), T, ptr(image_mask), obj_end,
                ptr(pack[0]) if pack is not None else 0, ptr(pack[1]) if pack is not None else 0,
                ptr(pack[4]) if (pack is not None and len(pack) > 4) else 0]

    def _layer_desc(self, ar, layer):
        key = ("bert_desc", id(layer))
        d = ar._views.get(key)
        if d is None:
            sa, so, li, lo = layer.attention.self, layer.attention.output, layer.intermediate, layer.output
            H3 = 3 * self.config.hidden_size
            c, g = ar.compute, ar.grad_view
            w = [c(sa.query.weight, H3).data_ptr(), c(so.dense.weight).data_ptr(), c(li.dense.weight).data_ptr(),
                 c(lo.dense.weight).data_ptr()]
            # (ptr, bytes) of the weights the neighbouring layers use first: the last product of a pass pulls them
            # towards the caches while it runs (MvltGemm.prefetch) -- next layer's qkv, previous layer's FFN-out
            layers = list(self.encoder.layer)
            i = next(k for k, l in enumerate(layers) if l is layer)
            nxt = c(layers[i + 1].attention.self.query.weight, H3) if i + 1 < len(layers) else None
            prv = c(layers[i - 1].output.dense.weight) if i > 0 else None
            for t in (nxt, prv):
                w += [t.data_ptr(), t.numel() * t.element_size()] if t is not None else [0, 0]
            f = [ar.master_span(sa.query.bias, H3).data_ptr(), so.dense.bias.data.data_ptr(), li.dense.bias.data.data_ptr(),
                 lo.dense.bias.data.data_ptr(), so.LayerNorm.weight.data.data_ptr(), so.LayerNorm.bias.data.data_ptr(),
                 lo.LayerNorm.weight.data.data_ptr(), lo.LayerNorm.bias.data.data_ptr()]
            gr = [g(sa.query.weight, H3).data_ptr(), g(sa.query.bias, H3).data_ptr(), g(so.dense.weight).data_ptr(),
                  g(so.dense.bias).data_ptr(), g(li.dense.weight).data_ptr(), g(li.dense.bias).data_ptr(),
                  g(lo.dense.weight).data_ptr(), g(lo.dense.bias).data_ptr(), g(so.LayerNorm.weight).data_ptr(),
                  g(so.LayerNorm.bias).data_ptr(), g(lo.LayerNorm.weight).data_ptr(), g(lo.LayerNorm.bias).data_ptr()]
            d = ar._views[key] = (w, f, gr)
        return d

    def _backward(self, sv, dhidden, dpooled):
        ar: Arena = sv["ar"]
        backward_begin(ar)
        lnq = ops.LnReduceQueue()
        g = ar.grad_view
        cfg = self.config
        B, Lq, H = sv["B"], sv["Lq"], cfg.hidden_size
        nH = cfg.num_attention_heads
        seed, p_h, p_a = sv["seed"], sv["p_h"], sv["p_a"]
        pack, rows = sv["pack"], sv["rows"]
        rd = pack[4] if (pack is not None and len(pack) > 4) else None
        if dhidden is None:                      # only the pooled output was used (VQA / retrieval heads)
            dx = torch.zeros((rows, H), dtype=sv["layers"][0][0].dtype, device=sv["layers"][0][0].device)
            dhidden = dx.new_empty(0)
        else:
            dx = dhidden.contiguous().view(rows, H)
        if dx.data_ptr() == dhidden.data_ptr() and self.pooler is not None and sv["pooled"] is not None:
            dx = dx.clone()                      # we accumulate the pooler gradient into it
        if self.pooler is not None and sv["pooled"] is not None and dpooled is not None:
            pd = self.pooler.dense
            dpre = ops.tanh_bwd(sv["pooled"], dpooled.contiguous())
            ops.gemm(dpre, sv["cls"], a_kmajor=True, b_kmajor=True, out=g(pd.weight), out_f32=True,
                     a_colsum=g(pd.bias))
            if pack is None:
                ops.gemm(dpre, ar.compute(pd.weight), b_kmajor=True, out=dx.view(B, Lq, H)[:, 0], accumulate=True)
            else:
                dx.index_add_(0, pack[3], ops.gemm(dpre, ar.compute(pd.weight), b_kmajor=True))
            ar.mark(pd.weight, pd.bias)
        native = sv.get("native", False)
        if native:
            hx, st, side = ops.host(), ops.stream_int(), ops.side_int(dx.device)
            akw = sv["akw"]
            attn_desc = self._attn_desc(sv["mode"], B, Lq, nH, akw["text_ids"], sv["T"], akw["image_mask"], akw["obj_end"], pack)
            sd = ops.s64(seed)
            for i in range(len(self.encoder.layer) - 1, -1, -1):
                layer = self.encoder.layer[i]
                sa, so, lo, li = layer.attention.self, layer.attention.output, layer.output, layer.intermediate
                w, f, gr = self._layer_desc(ar, layer)
                dx = hx.bert_layer_bwd(dx, sv["layers"][i], w, f, gr, H, cfg.intermediate_size, attn_desc, p_h, p_a, sd, i,
                                       st, side)
                sv["layers"][i] = None
                ar.mark(lo.LayerNorm.weight, lo.LayerNorm.bias, lo.dense.weight, lo.dense.bias, li.dense.weight,
                        li.dense.bias, so.LayerNorm.weight, so.LayerNorm.bias, so.dense.weight, so.dense.bias,
                        sa.query.weight, sa.key.weight, sa.value.weight, sa.query.bias, sa.key.bias, sa.value.bias)
            hx.lnq_flush(st)
        for i in range(len(self.encoder.layer) - 1, -1, -1) if not native else ():
            layer = self.encoder.layer[i]
            sa, so = layer.attention.self, layer.attention.output
            (x, qkv, ctx, lse, y1, m1, r1, x1, h, a, y2, m2, r2) = sv["layers"][i]
            lo, li = layer.output, layer.intermediate
            if p_h > 0:
                dy2, dz2 = ops.layernorm_bwd(dx, y2, m2, r2, lo.LayerNorm.weight.data, g(lo.LayerNorm.weight),
                                             g(lo.LayerNorm.bias), branch=dict(dropout=(p_h, seed, 8 * i + 2)), defer=lnq,
                                             rows_dev=rd)
            else:
                dy2 = dz2 = ops.layernorm_bwd(dx, y2, m2, r2, lo.LayerNorm.weight.data, g(lo.LayerNorm.weight),
                                              g(lo.LayerNorm.bias), defer=lnq, rows_dev=rd)
            dh = ops.gemm(dz2, ar.compute(lo.dense.weight), b_kmajor=True, mul_gelu_grad=h, m_dev=rd)
            dx1 = ops.gemm(dh, ar.compute(li.dense.weight), b_kmajor=True, residual=dy2, m_dev=rd)
            if p_h > 0:
                dy1, dz1 = ops.layernorm_bwd(dx1, y1, m1, r1, so.LayerNorm.weight.data, g(so.LayerNorm.weight),
                                             g(so.LayerNorm.bias), branch=dict(dropout=(p_h, seed, 8 * i + 1)), defer=lnq,
                                             rows_dev=rd)
            else:
                dy1 = dz1 = ops.layernorm_bwd(dx1, y1, m1, r1, so.LayerNorm.weight.data, g(so.LayerNorm.weight),
                                              g(so.LayerNorm.bias), defer=lnq, rows_dev=rd)
            dctx = ops.gemm(dz1, ar.compute(so.dense.weight), b_kmajor=True, m_dev=rd)
            dqkv = ops.attn_bwd(dctx, qkv, ctx, lse, sv["mode"], B, Lq, nH, H // nH, (H // nH) ** -0.5,
                                dropout=(p_a, seed, 8 * i + 0), **sv["akw"])
            dx_in = ops.gemm(dqkv, ar.compute(sa.query.weight, 3 * H), b_kmajor=True, residual=dy1, m_dev=rd)
            # weight / bias gradients on the side stream (off the critical path)
            xtra = () if rd is None else (rd,)
            with ops.on_side(dx.device, dz2, a, dh, x1, dz1, ctx, dqkv, x):
                ops.wgrad_group([(dz2, a, g(lo.dense.weight), g(lo.dense.bias)) + xtra,
                                 (dh, x1, g(li.dense.weight), g(li.dense.bias)) + xtra,
                                 (dz1, ctx, g(so.dense.weight), g(so.dense.bias)) + xtra,
                                 (dqkv, x, g(sa.query.weight, 3 * H), g(sa.query.bias, 3 * H)) + xtra])
            dx = dx_in
            ar.mark(lo.LayerNorm.weight, lo.LayerNorm.bias, lo.dense.weight, lo.dense.bias, li.dense.weight,
                    li.dense.bias, so.LayerNorm.weight, so.LayerNorm.bias, so.dense.weight, so.dense.bias,
                    sa.query.weight, sa.key.weight, sa.value.weight, sa.query.bias, sa.key.bias, sa.value.bias)
        lnq.flush()
        # ---- embeddings: dense f32 table gradients, like nn.Embedding in the reference
        we, pe, te = self.word_embeddings.weight, self.position_embeddings.weight, self.token_type_embeddings.weight
        g(we).zero_()          # scatter-add target; the position / type tables are overwritten whole by mvlt_embed_bwd
        dimg = ops.embed_bwd(dx.view(B, Lq, H) if pack is None else dx, sv["text_idx"], sv["n_img"], we.data, pe.data,
                             te.data, cfg.cls_token_id, cfg.sep_token_id, g(we), g(pe), g(te), pack=pack, B=B)
        ops.join_side(dx.device)
        ar.mark(we, pe, te)
        return dimg
