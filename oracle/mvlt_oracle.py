"""CPU oracle for the MVLT vision-language hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain fp32 PyTorch *restatement* (functional, state-dict driven)
of the arithmetic the reference performs on the hot path.  It is the checker:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  Nothing under ``medical-vision-langauge-transformer_amd/``
imports it and the product path never falls back to it.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real
reference (``/root/reference``, with dependency shims) in the build container,
loads the same formula weights, and stores its outputs under ``tests/golden``;
``tests/test_oracle_golden.py`` checks this file against those vectors.  The
HF ``transformers`` BERT blocks the reference imports (``modules/model.py:4-5``)
are third-party (pinned ``>=4.16.0`` in the reference README); their arithmetic
is restated here from the installed 5.15.0 source
(``transformers/models/bert/modeling_bert.py:111-136,164-203,282-293,325-351,
451-463,466-506``) and pinned through the same golden vectors.

Exception -- ``beam_decode_recompute``: PARITY UNPINNED.  The reference delegates the beam
bookkeeping to HF ``BeamSearchScorer`` (third-party, ``transformers>=4.16.0``, absent from the
installed release, so the reference's beam search cannot run here); it is restated from the 4.16
semantics and only pins the build's cached implementation against an independent statement of the
same algorithm.  ``greedy_decode_recompute`` is pinned through the golden-pinned forward it calls
(the reference's own cached loop does not run under the installed HF either).

All functions take ``sd`` -- a dict of tensors with the *reference's* state-dict
key names -- plus a key prefix, so the HIP modules' ``state_dict()`` can be fed
to the oracle unchanged.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------
# configuration (values: modules/swin_small_patch4_window7_224.yaml:1-8,
# modules/swin_transformer_config.py:58-76, HF BertConfig defaults,
# modules/config.py:4-50)
# --------------------------------------------------------------------------
@dataclass
class SwinCfg:
    img_size: int = 224
    patch_size: int = 4
    in_chans: int = 3
    embed_dim: int = 96
    depths: Sequence[int] = (2, 2, 18, 2)
    num_heads: Sequence[int] = (3, 6, 12, 24)
    window_size: int = 7
    mlp_ratio: float = 4.0
    drop_path_rate: float = 0.3
    num_classes: int = 1000

    @property
    def res0(self) -> int:
        return self.img_size // self.patch_size

    @property
    def num_features(self) -> int:
        return self.embed_dim * 2 ** (len(self.depths) - 1)


@dataclass
class BertCfg:
    vocab_size: int = 30522
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    max_position_embeddings: int = 512
    type_vocab_size: int = 3
    layer_norm_eps: float = 1e-12
    hidden_dropout_prob: float = 0.1
    attention_probs_dropout_prob: float = 0.1
    cls_token_id: int = 101
    sep_token_id: int = 102
    mask_token_id: int = 103
    eos_token_id: int = 104
    result_num: int = 224


class Dropper:
    """Dropout hook.  ``mode='off'`` = eval.  ``mode='torch'`` = torch RNG.
    ``mode='given'``: masks supplied by the test (e.g. dumped from the HIP
    kernels' counter RNG) keyed by tag, so train-mode parity is exact."""

    def __init__(self, mode: str = "off", masks: Optional[Dict[str, Tensor]] = None):
        self.mode = mode
        self.masks = masks or {}

    def __call__(self, x: Tensor, p: float, tag: str) -> Tensor:
        if self.mode == "off" or p == 0.0:
            return x
        if self.mode == "torch":
            return F.dropout(x, p, True)
        keep = self.masks[tag].to(x.dtype).reshape(x.shape)
        return x * keep / (1.0 - p)

    def path(self, x: Tensor, p: float, tag: str) -> Tensor:
        """Per-sample stochastic depth (timm DropPath, imported at
        visual_feature_extractor.py:122, used :384-385)."""
        if self.mode == "off" or p == 0.0:
            return x
        if self.mode == "torch":
            keep = (torch.rand(x.shape[0], device=x.device) >= p).to(x.dtype)
        else:
            keep = self.masks[tag].to(x.dtype).reshape(-1)
        shape = (x.shape[0],) + (1,) * (x.dim() - 1)
        return x * (keep / (1.0 - p)).view(shape)


EVAL = Dropper("off")


# --------------------------------------------------------------------------
# INT tables (bit-exact rows a2-a5 of SURVEY.md section 8)
# --------------------------------------------------------------------------
def relative_position_index(ws: int) -> Tensor:
    """idx[i,j] = (y_i-y_j+ws-1)*(2ws-1) + (x_i-x_j+ws-1)
    (visual_feature_extractor.py:203-213)."""
    ys, xs = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")
    y = ys.reshape(-1)
    x = xs.reshape(-1)
    dy = y[:, None] - y[None, :] + ws - 1
    dx = x[:, None] - x[None, :] + ws - 1
    return dy * (2 * ws - 1) + dx


def shift_region_ids(H: int, W: int, ws: int, shift: int) -> Tensor:
    """Region id 0..8 per (h, w) of the *shifted* image
    (visual_feature_extractor.py:321-339)."""
    def band(n):
        b = torch.zeros(n, dtype=torch.int64)
        b[n - ws:n - shift] = 1
        b[n - shift:] = 2
        return b
    return band(H)[:, None] * 3 + band(W)[None, :]


def window_token_map(H: int, W: int, ws: int, shift: int) -> Tensor:
    """src[w*ws*ws + s] = token index (h*W + w) of the un-shifted image that
    lands in window w, slot s after roll(-shift) + window_partition
    (visual_feature_extractor.py:144-156, :360-367)."""
    hh, ww = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    src_h = (hh + shift) % H          # roll(-shift): out[h] = in[(h+shift)%H]
    src_w = (ww + shift) % W
    src = src_h * W + src_w           # [H, W] in shifted coordinates
    src = src.view(H // ws, ws, W // ws, ws).permute(0, 2, 1, 3).reshape(-1)
    return src


def shift_attn_mask(H: int, W: int, ws: int, shift: int) -> Tensor:
    """[nW, ws*ws, ws*ws] with 0 / -100.0 (visual_feature_extractor.py:341-344)."""
    reg = shift_region_ids(H, W, ws, shift)
    reg = reg.view(H // ws, ws, W // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    diff = reg[:, None, :] - reg[:, :, None]
    return torch.where(diff != 0, torch.tensor(-100.0), torch.tensor(0.0))


def patch_merge_map(H: int, W: int) -> Tensor:
    """[H/2*W/2, 4] source tokens in concat order x0,x1,x2,x3
    (visual_feature_extractor.py:435-439)."""
    i, j = torch.meshgrid(torch.arange(H // 2), torch.arange(W // 2), indexing="ij")
    i = i.reshape(-1)
    j = j.reshape(-1)
    return torch.stack([(2 * i) * W + 2 * j, (2 * i + 1) * W + 2 * j,
                        (2 * i) * W + 2 * j + 1, (2 * i + 1) * W + 2 * j + 1], 1)


def vl_layout(T: int, n_img: int) -> Dict[str, Tensor]:
    """Sequence layout of model.py:110-160: [CLS] img*n [SEP] text*T."""
    L = n_img + 2 + T
    pos = torch.arange(L)
    obj_end = n_img + 1
    return {"position_ids": pos, "token_type_ids": (pos <= obj_end).long(),
            "obj_end": torch.tensor(obj_end), "text_end": torch.tensor(obj_end + 1 + T)}


def seq2seq_bool_mask(L: int, obj_end: int) -> Tensor:
    """(col<=row) OR (col<=obj_end)  (model.py:118-123)."""
    r = torch.arange(L)[:, None]
    c = torch.arange(L)[None, :]
    return (c <= r) | (c <= obj_end)


def bidir_bool_mask(text_idx: Optional[Tensor], B: int, n_img: int,
                    image_mask: Optional[Tensor] = None) -> Tensor:
    """cat(1, image_mask, 1, text_idx>0)  (model.py:125-128)."""
    one = torch.ones(B, 1, dtype=torch.bool)
    im = torch.ones(B, n_img, dtype=torch.bool) if image_mask is None else image_mask.bool()
    parts = [one, im, one]
    if text_idx is not None:
        parts.append(text_idx > 0)
    return torch.cat(parts, 1)


def cached_step_rows(past: int, n_new: int = 2) -> Tuple[Tensor, Tensor]:
    """KV-cache step (model.py:82-108): position ids and the last 2 rows of a
    causal (past+n_new)^2 mask."""
    tot = past + n_new
    pos = torch.arange(past, tot)
    r = torch.arange(tot)[:, None]
    c = torch.arange(tot)[None, :]
    return pos, (c <= r)[-2:]


# --------------------------------------------------------------------------
# Swin (visual_feature_extractor.py:125-701)
# --------------------------------------------------------------------------
def _ln(x: Tensor, sd: SD, p: str, eps: float = 1e-5) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def _lin(x: Tensor, sd: SD, p: str) -> Tensor:
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def window_attention(x: Tensor, sd: SD, p: str, nH: int, ws: int,
                     mask: Optional[Tensor]) -> Tensor:
    """visual_feature_extractor.py:224-254.  x: [B_, N, C]."""
    B_, N, C = x.shape
    hd = C // nH
    qkv = _lin(x, sd, p + ".qkv").view(B_, N, 3, nH, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
    att = q @ k.transpose(-2, -1)
    idx = relative_position_index(ws).to(x.device)
    bias = sd[p + ".relative_position_bias_table"][idx.view(-1)].view(N, N, nH).permute(2, 0, 1)
    att = att + bias[None]
    if mask is not None:
        nW = mask.shape[0]
        att = (att.view(B_ // nW, nW, nH, N, N) + mask[None, :, None]).view(-1, nH, N, N)
    att = att.softmax(-1)
    out = (att @ v).transpose(1, 2).reshape(B_, N, C)
    return _lin(out, sd, p + ".proj")


def swin_block(x: Tensor, sd: SD, p: str, H: int, W: int, nH: int, ws: int, shift: int,
               dp: float, drop: Dropper) -> Tensor:
    """visual_feature_extractor.py:350-387."""
    B, L, C = x.shape
    if min(H, W) <= ws:            # :302-305
        shift, ws_eff = 0, min(H, W)
    else:
        ws_eff = ws
    src = window_token_map(H, W, ws_eff, shift).to(x.device)
    xn = _ln(x, sd, p + ".norm1")
    xw = xn[:, src].reshape(-1, ws_eff * ws_eff, C)
    mask = shift_attn_mask(H, W, ws_eff, shift).to(x.device) if shift > 0 else None
    aw = window_attention(xw, sd, p + ".attn", nH, ws_eff, mask).reshape(B, L, C)
    a = torch.empty_like(aw)
    a[:, src] = aw                  # window_reverse + roll(+shift)
    x = x + drop.path(a, dp, p + ".dp1")
    h = F.gelu(_lin(_ln(x, sd, p + ".norm2"), sd, p + ".mlp.fc1"))
    return x + drop.path(_lin(h, sd, p + ".mlp.fc2"), dp, p + ".dp2")


def patch_merging(x: Tensor, sd: SD, p: str, H: int, W: int) -> Tensor:
    """visual_feature_extractor.py:424-445."""
    B, L, C = x.shape
    x = x[:, patch_merge_map(H, W).to(x.device).reshape(-1)].reshape(B, L // 4, 4 * C)
    return F.linear(_ln(x, sd, p + ".norm"), sd[p + ".reduction.weight"])


def patch_embed(img: Tensor, sd: SD, p: str, ps: int) -> Tensor:
    """visual_feature_extractor.py:557-565."""
    x = F.conv2d(img, sd[p + ".proj.weight"], sd[p + ".proj.bias"], stride=ps)
    return _ln(x.flatten(2).transpose(1, 2), sd, p + ".norm")


def swin_forward(img: Tensor, sd: SD, p: str, cfg: SwinCfg, drop: Dropper = EVAL,
                 taps: Optional[dict] = None) -> Tensor:
    """SwinTransformer.forward_features (visual_feature_extractor.py:676-693):
    returns un-pooled tokens [B, 49, 768]."""
    x = patch_embed(img, sd, p + "patch_embed", cfg.patch_size)
    if taps is not None:
        taps["patch_embed"] = x
    nblk = sum(cfg.depths)
    dpr = torch.linspace(0, cfg.drop_path_rate, nblk).tolist()      # :633
    H = W = cfg.res0
    bi = 0
    for s, depth in enumerate(cfg.depths):
        for j in range(depth):
            x = swin_block(x, sd, f"{p}layers.{s}.blocks.{j}", H, W, cfg.num_heads[s],
                           cfg.window_size, 0 if j % 2 == 0 else cfg.window_size // 2,
                           dpr[bi], drop)
            bi += 1
        if s < len(cfg.depths) - 1:
            x = patch_merging(x, sd, f"{p}layers.{s}.downsample", H, W)
            H, W = H // 2, W // 2
        if taps is not None:
            taps[f"stage{s}"] = x
    return _ln(x, sd, p + "norm")


def conv_layer(v: Tensor, sd: SD, cfg: SwinCfg, drop: Dropper = EVAL, p: str = "conv.") -> Tensor:
    """Conv_layer.forward (model.py:238-266): Swin -> GELU; 5-D input = two passes."""
    f = lambda im: F.gelu(swin_forward(im, sd, p + "conv.0.", cfg, drop))
    if v.dim() == 5:
        return torch.cat([f(v[:, 0]), f(v[:, 1])], 1)
    return f(v)


# --------------------------------------------------------------------------
# MVLBert (model.py:16-183) + HF BERT layers
# --------------------------------------------------------------------------
def vl_embeddings(sd: SD, p: str, cfg: BertCfg, text_idx: Optional[Tensor], img: Tensor) -> Tensor:
    """model.py:133-158 -- note: NO LayerNorm / dropout on the sum."""
    B, n_img, _ = img.shape
    we = sd[p + "word_embeddings.weight"]
    parts = [we[cfg.cls_token_id].expand(B, 1, -1), img, we[cfg.sep_token_id].expand(B, 1, -1)]
    T = 0
    if text_idx is not None:
        parts.append(we[text_idx])
        T = text_idx.shape[1]
    lay = vl_layout(T, n_img)
    x = torch.cat(parts, 1)
    x = x + sd[p + "token_type_embeddings.weight"][lay["token_type_ids"].to(img.device)][None]
    return x + sd[p + "position_embeddings.weight"][lay["position_ids"].to(img.device)][None]


def bert_layer(x: Tensor, add_mask: Optional[Tensor], sd: SD, p: str, cfg: BertCfg, drop: Dropper,
               past: Optional[Tuple[Tensor, Tensor]] = None):
    """One HF BertLayer (modeling_bert.py:164-203, 282-293, 325-351).
    add_mask broadcastable to [B, nH, Lq, Lk].  Returns (out, (k, v))."""
    B, Lq, Hd = x.shape
    nH = cfg.num_attention_heads
    hd = Hd // nH
    split = lambda t: t.view(B, -1, nH, hd).transpose(1, 2)
    q = split(_lin(x, sd, p + ".attention.self.query"))
    k = split(_lin(x, sd, p + ".attention.self.key"))
    v = split(_lin(x, sd, p + ".attention.self.value"))
    if past is not None:
        k = torch.cat([past[0], k], 2)
        v = torch.cat([past[1], v], 2)
    att = (q @ k.transpose(-1, -2)) * hd ** -0.5
    if add_mask is not None:
        att = att + add_mask
    att = drop(att.softmax(-1), cfg.attention_probs_dropout_prob, p + ".attn_drop")
    ctx = (att @ v).transpose(1, 2).reshape(B, Lq, Hd)
    h = drop(_lin(ctx, sd, p + ".attention.output.dense"), cfg.hidden_dropout_prob, p + ".drop1")
    x1 = _ln(h + x, sd, p + ".attention.output.LayerNorm", cfg.layer_norm_eps)
    inter = F.gelu(_lin(x1, sd, p + ".intermediate.dense"))
    h2 = drop(_lin(inter, sd, p + ".output.dense"), cfg.hidden_dropout_prob, p + ".drop2")
    return _ln(h2 + x1, sd, p + ".output.LayerNorm", cfg.layer_norm_eps), (k, v)


def additive_mask(bool_mask: Tensor) -> Tensor:
    """get_extended_attention_mask (model.py:162-183): (1-m)*-10000."""
    m = bool_mask.to(torch.float32)
    m = m[:, None, None, :] if m.dim() == 2 else m[:, None, :, :]
    return (1.0 - m) * -10000.0


def mvlbert_forward(sd: SD, cfg: BertCfg, text_idx: Optional[Tensor], img: Tensor,
                    seq2seq: bool, drop: Dropper = EVAL, p: str = "MVLBert.",
                    image_mask: Optional[Tensor] = None, return_kv: bool = False):
    """MVLBert.forward without cache (model.py:35-72, :110-160).
    Returns dict(hidden, pooled, text, image, sep[, kv])."""
    B, n_img, _ = img.shape
    T = 0 if text_idx is None else text_idx.shape[1]
    L = n_img + 2 + T
    obj_end = n_img + 1
    x = vl_embeddings(sd, p, cfg, text_idx, img)
    if seq2seq:
        bm = seq2seq_bool_mask(L, obj_end)[None].expand(B, L, L)
    else:
        bm = bidir_bool_mask(text_idx.cpu() if text_idx is not None else None, B, n_img, image_mask)
    am = additive_mask(bm.to(img.device))
    kv = []
    for i in range(cfg.num_hidden_layers):
        x, kvi = bert_layer(x, am, sd, f"{p}encoder.layer.{i}", cfg, drop)
        kv.append(kvi)
    out = {"hidden": x, "image": x[:, 1:obj_end], "text": x[:, obj_end + 1:obj_end + 1 + T],
           "sep": x[:, obj_end], "pooled": None}
    if p + "pooler.dense.weight" in sd:
        out["pooled"] = torch.tanh(_lin(x[:, 0], sd, p + "pooler.dense"))   # modeling_bert.py:451-463
    if return_kv:
        out["kv"] = kv
    return out


def mvlbert_cached_step(sd: SD, cfg: BertCfg, new_ids: Tensor, kv: List[Tuple[Tensor, Tensor]],
                        p: str = "MVLBert."):
    """KV-cache branch (model.py:82-108): new_ids [B,2] = [last_tok, MASK]."""
    past = kv[0][0].shape[2]
    pos, rows = cached_step_rows(past, new_ids.shape[1])
    x = (sd[p + "word_embeddings.weight"][new_ids]
         + sd[p + "token_type_embeddings.weight"][0][None, None]
         + sd[p + "position_embeddings.weight"][pos][None])
    am = additive_mask(rows[None].expand(new_ids.shape[0], -1, -1))
    new_kv = []
    for i in range(cfg.num_hidden_layers):
        x, kvi = bert_layer(x, am, sd, f"{p}encoder.layer.{i}", cfg, EVAL, past=kv[i])
        new_kv.append(kvi)
    return x, new_kv


def mlm_head(x: Tensor, sd: SD, p: str, cfg: BertCfg) -> Tensor:
    """BertOnlyMLMHead (modeling_bert.py:466-506)."""
    h = _ln(F.gelu(_lin(x, sd, p + ".predictions.transform.dense")), sd,
            p + ".predictions.transform.LayerNorm", cfg.layer_norm_eps)
    bias = sd.get(p + ".predictions.decoder.bias", sd.get(p + ".predictions.bias"))
    return F.linear(h, sd[p + ".predictions.decoder.weight"], bias)


def pretrain_loss(sd: SD, scfg: SwinCfg, bcfg: BertCfg, image: Tensor, caption_masked: Tensor,
                  caption_label: Tensor, itm_label: Optional[Tensor], seq2seq: bool,
                  itm_task: bool = False, drop: Dropper = EVAL, taps: Optional[dict] = None) -> Tensor:
    """MVLBertForPretraining.forward (model.py:372-420).  The host coin flip
    (model.py:390-394) is an explicit argument."""
    feat = conv_layer(image, sd, scfg, drop)
    o = mvlbert_forward(sd, bcfg, caption_masked, feat, seq2seq, drop)
    head = "MLM_head_seq2seq" if seq2seq else "MLM_head_bidir"
    logits = mlm_head(o["text"], sd, head, bcfg)
    loss = F.cross_entropy(logits.transpose(1, 2), caption_label, ignore_index=-100)
    if taps is not None:
        taps.update(feat=feat, hidden=o["hidden"], logits=logits, pooled=o["pooled"])
    if itm_task:
        loss = loss.mean() + F.cross_entropy(_lin(o["pooled"], sd, "ITM_mlp"), itm_label).mean()
    return loss


def vqa_forward(sd: SD, scfg: SwinCfg, bcfg: BertCfg, image: Tensor, question: Tensor,
                drop: Dropper = EVAL):
    """MVLBertForVQA.forward (model.py:329-349) -> (prob, logits)."""
    feat = conv_layer(image, sd, scfg, drop)
    o = mvlbert_forward(sd, bcfg, question, feat, False, drop)
    logits = _lin(drop(o["pooled"], bcfg.hidden_dropout_prob, "vqa.drop"), sd, "final_mlp.1")
    return logits.softmax(-1), logits


def greedy_decode_recompute(sd: SD, scfg: SwinCfg, bcfg: BertCfg, image: Tensor, max_len: int):
    """Oracle for greedy_search (model.py:826-984) by FULL-SEQUENCE RECOMPUTE:
    step t runs [CLS] img [SEP] y_1..y_{t-1} [MASK] with the seq2seq mask and
    reads the last position; equals the cached 2-token step because the mask
    is causal over text and the cache drops the [MASK] slot (model.py:890-894)."""
    feat = conv_layer(image, sd, scfg)
    B = image.shape[0]
    ids = torch.zeros(B, 0, dtype=torch.long)
    unfinished = torch.ones(B, dtype=torch.long)
    for _ in range(max_len):
        inp = torch.cat([ids, torch.full((B, 1), bcfg.mask_token_id)], 1)
        o = mvlbert_forward(sd, bcfg, inp, feat, True)
        nxt = mlm_head(o["hidden"][:, -1], sd, "MLM_head_seq2seq", bcfg).argmax(-1)
        nxt = nxt * unfinished + 0 * (1 - unfinished)          # pad_token_id = 0
        ids = torch.cat([ids, nxt[:, None]], 1)
        unfinished = unfinished * (nxt != bcfg.eos_token_id).long()
        if unfinished.max() == 0:
            break
    return ids


class _HypsHF416:
    """transformers 4.16 ``BeamHypotheses`` restated (tensor hypotheses, as HF keeps them)."""

    def __init__(self, num_beams, length_penalty=1.0, early_stopping=False):
        self.num_beams, self.length_penalty, self.early_stopping = num_beams, length_penalty, early_stopping
        self.beams, self.worst_score = [], 1e9

    def add(self, hyp, sum_logprobs):
        score = sum_logprobs / (hyp.shape[-1] ** self.length_penalty)
        if len(self.beams) < self.num_beams or score > self.worst_score:
            self.beams.append((score, hyp))
            if len(self.beams) > self.num_beams:
                srt = sorted([(sc, i) for i, (sc, _) in enumerate(self.beams)])
                del self.beams[srt[0][1]]
                self.worst_score = srt[1][0]
            else:
                self.worst_score = min(score, self.worst_score)

    def is_done(self, best_sum_logprobs, cur_len):
        if len(self.beams) < self.num_beams:
            return False
        if self.early_stopping:
            return True
        return self.worst_score >= best_sum_logprobs / cur_len ** self.length_penalty


def beam_decode_recompute(sd: SD, scfg: SwinCfg, bcfg: BertCfg, image: Tensor, num_beams: int, max_len: int):
    """Oracle for ``beam_search`` (model.py:636-816) by FULL-SEQUENCE RECOMPUTE per step (no cache), with the
    scorer bookkeeping of transformers 4.16 ``BeamSearchScorer.process/finalize`` written out inline (the class is
    third-party and absent from the installed transformers: PARITY WITH THE REFERENCE IS UNPINNED; this pins the
    build's cached implementation against an independent statement of the same algorithm)."""
    feat0 = conv_layer(image, sd, scfg)
    B, nb = image.shape[0], num_beams
    feat = feat0.repeat_interleave(nb, dim=0)                                   # _expand_inputs_for_generation
    pad, eos, mask = 0, bcfg.eos_token_id, bcfg.mask_token_id
    hyps = [_HypsHF416(nb) for _ in range(B)]
    done = [False] * B
    beam_scores = torch.zeros(B, nb)
    beam_scores[:, 1:] = -1e9
    beam_scores = beam_scores.view(-1)
    gen = None                                                                   # generated ids [B*nb, cur_len]
    input_ids = torch.full((B * nb, 1), mask)                                    # step 0: the [MASK] column (:701-702)
    cur_len = 0
    V = bcfg.vocab_size
    while cur_len < max_len:
        inp = input_ids if gen is None else torch.cat([gen, torch.full((B * nb, 1), mask)], 1)
        o = mvlbert_forward(sd, bcfg, inp, feat, True)
        logits = mlm_head(o["hidden"][:, -1], sd, "MLM_head_seq2seq", bcfg)
        scores = torch.log_softmax(logits, -1) + beam_scores[:, None]
        scores = scores.view(B, nb * V)
        nscores, ntok = torch.topk(scores, 2 * nb, dim=1, largest=True, sorted=True)
        nidx = torch.div(ntok, V, rounding_mode="floor")
        ntok = ntok % V
        # ---- BeamSearchScorer.process
        ids_for_scorer = input_ids if gen is None else gen
        slen = ids_for_scorer.shape[-1]
        nbs, nbt, nbi = torch.zeros(B, nb), torch.zeros(B, nb, dtype=torch.long), torch.zeros(B, nb, dtype=torch.long)
        for b in range(B):
            if done[b]:
                nbs[b] = 0; nbt[b] = pad; nbi[b] = 0
                continue
            k = 0
            for rank in range(2 * nb):
                t, sc, ix = int(ntok[b, rank]), float(nscores[b, rank]), int(nidx[b, rank])
                row = b * nb + ix
                if t == eos:
                    if rank >= nb:
                        continue
                    hyps[b].add(ids_for_scorer[row].clone(), sc)
                else:
                    nbs[b, k], nbt[b, k], nbi[b, k] = sc, t, row
                    k += 1
                if k == nb:
                    break
            assert k == nb
            done[b] = done[b] or hyps[b].is_done(float(nscores[b].max()), slen)
        beam_scores, btok, bidx = nbs.view(-1), nbt.view(-1), nbi.view(-1)
        gen = btok[:, None] if cur_len == 0 else torch.cat([gen[bidx], btok[:, None]], -1)
        cur_len += 1
        if all(done):
            break
    # ---- BeamSearchScorer.finalize (num_beam_hyps_to_keep = 1, max_length = the model config's)
    for b in range(B):
        if done[b]:
            continue
        for k in range(nb):
            hyps[b].add(gen[b * nb + k], float(beam_scores[b * nb + k]))
    best = [sorted(h.beams, key=lambda x: x[0]).pop()[1] for h in hyps]
    lens = [len(h) for h in best]
    width = min(max(lens) + 1, max_len)
    out = torch.full((B, width), pad, dtype=torch.long)
    for i, h in enumerate(best):
        out[i, :lens[i]] = h[:width]
        if lens[i] < max_len:
            out[i, lens[i]] = eos
    return out


# --------------------------------------------------------------------------
# deterministic formula weights shared by the golden generator and the tests
# --------------------------------------------------------------------------
def formula_fill(named_shapes: Sequence[Tuple[str, Tuple[int, ...], torch.dtype]]) -> SD:
    """k-th float entry: scale_k * sin(0.37*arange(n) + k); LayerNorm weights
    1 + that.  Integer buffers are skipped (left to the module)."""
    out: SD = {}
    k = 0
    for name, shape, dtype in named_shapes:
        if not dtype.is_floating_point:
            continue
        n = int(math.prod(shape)) if len(shape) else 1
        base = torch.sin(0.37 * torch.arange(n, dtype=torch.float64) + k).to(torch.float32).view(shape)
        low = name.lower()
        if "attn_mask" in low:
            k += 1
            continue
        if ("norm" in low) and name.endswith("weight"):
            t = 1.0 + 0.1 * base
        elif name.endswith("bias") and "relative_position" not in name:
            t = 0.02 * base
        elif "embeddings.weight" in name:
            t = 0.05 * base
        elif "relative_position_bias_table" in name:
            t = 0.2 * base
        else:
            fan_in = shape[-1] if len(shape) == 2 else int(math.prod(shape[1:])) if len(shape) > 1 else n
            t = base * (1.0 / math.sqrt(fan_in))
        out[name] = t
        k += 1
    return out


def hash_fill(named_shapes: Sequence[Tuple[str, Tuple[int, ...], torch.dtype]], seed: int = 20261003) -> SD:
    """Well-conditioned deterministic weights both sides can regenerate: element i of the k-th float entry is an
    integer hash of (k, i) mapped to U[-1, 1) (exact integer arithmetic, identical on every platform), scaled like a
    trained network: Linear / conv weights U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (torch's own default), LayerNorm
    weights 1 +- 0.1, biases +-0.02, embeddings +-0.05, relative-position tables +-0.2.  Unlike the sin() formula
    of formula_fill (every matrix has rank 2) these matrices are full rank, so bf16 errors are not amplified by the
    fixture itself."""
    out: SD = {}
    k = 0
    M32 = 0xFFFFFFFF
    for name, shape, dtype in named_shapes:
        if not dtype.is_floating_point:
            continue
        low = name.lower()
        if "attn_mask" in low:
            k += 1
            continue
        n = int(math.prod(shape)) if len(shape) else 1
        h = (torch.arange(n, dtype=torch.int64) * 2654435761 + (k + 1) * 40503 + seed) & M32
        for _ in range(2):
            h = h ^ (h >> 16)
            h = (h * 0x45D9F3B) & M32
        h = h ^ (h >> 16)
        u = (h.to(torch.float64) / 2147483648.0 - 1.0).to(torch.float32).view(shape)
        if ("norm" in low) and name.endswith("weight"):
            t = 1.0 + 0.1 * u
        elif name.endswith("bias") and "relative_position" not in name:
            t = 0.02 * u
        elif "embeddings.weight" in name:
            t = 0.05 * u
        elif "relative_position_bias_table" in name:
            t = 0.2 * u
        else:
            fan_in = shape[-1] if len(shape) == 2 else int(math.prod(shape[1:])) if len(shape) > 1 else n
            t = u * (1.0 / math.sqrt(fan_in))
        out[name] = t
        k += 1
    return out

