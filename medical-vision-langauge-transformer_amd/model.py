"""Task heads and wrappers, drop-in for reference ``modules/model.py:186-546`` and
``modules/config.py`` (same class names, constructor / forward signatures,
state-dict keys): Conv_layer, MVLBertForPretraining, MVLBertForVQA,
MVLBertForRetrieval, MVLBertForImageCaption, and the MVLBertConfig family.
"""
from __future__ import annotations

import json
import os
import random
from typing import Optional

import torch
import torch.nn as nn

from . import ops
from .arena import Arena
from .bert import MVLBert
from .runtime import GraphedEval, backward_begin, compute_dtype_of, next_seed
from .swin import SwinTransformer


# ----------------------------------------------------------------------------- configs (modules/config.py:4-72)
class MVLBertConfig:
    """Plain restatement of ``MVLBertConfig(BertConfig)``: bert-base-uncased
    defaults plus the reference's extra fields (modules/config.py:4-27)."""

    def __init__(self, **kwargs):
        self.vocab_size = 30522
        self.hidden_size = 768
        self.num_hidden_layers = 12
        self.num_attention_heads = 12
        self.intermediate_size = 3072
        self.hidden_act = "gelu"
        self.max_position_embeddings = 512
        self.layer_norm_eps = 1e-12
        self.initializer_range = 0.02
        self.is_decoder = False
        self.pad_token_id = 0
        self.type_vocab_size = 3
        self.MLM_task = True
        self.ITM_task = True
        self.conv = 'swintransformer'
        self.result_num = 224
        self.lr = 4e-5
        self.max_length = 40
        self.eos_token_id, self.cls_token_id, self.sep_token_id, self.mask_token_id = 104, 101, 102, 103
        self.attention_probs_dropout_prob = 0.0
        self.hidden_dropout_prob = 0.0
        # Swin-S (modules/swin_small_patch4_window7_224.yaml:1-8 + swin_transformer_config.py:58-76)
        self.swin = dict(img_size=224, patch_size=4, in_chans=3, num_classes=1000, embed_dim=96,
                         depths=[2, 2, 18, 2], num_heads=[3, 6, 12, 24], window_size=7, mlp_ratio=4.,
                         qkv_bias=True, qk_scale=None, drop_rate=0.0, drop_path_rate=0.3, ape=False,
                         patch_norm=True, use_checkpoint=False)
        for k, v in kwargs.items():
            setattr(self, k, v)

    def use_swin_base(self, drop_path_rate=0.5):
        """Swin-B (swin_base_patch4_window7_224: embed 128, depths 2/2/18/2, heads 4/8/16/32, drop path 0.5).
        Its tokens are 1024-d: the reference has no way to feed them to the 768-d encoder, so this
        build adds one Linear(1024, hidden_size) after the GELU (``conv.feature_proj``, BASELINE config #5)."""
        self.swin.update(embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], drop_path_rate=drop_path_rate)
        self.swin_feature_proj = True
        return self

    def update_special_tokens(self, tokenizer):
        self.eos_token_id, self.cls_token_id, self.sep_token_id, self.mask_token_id = \
            tokenizer.convert_tokens_to_ids(['[END]', '[CLS]', '[SEP]', '[MASK]'])
        self.vocab_size = len(tokenizer)

    def to_dict(self):
        return {k: v for k, v in self.__dict__.items() if isinstance(v, (int, float, str, bool, list, dict, type(None)))}

    @classmethod
    def from_pretrained(cls, path, **kwargs):
        cfg = cls(**kwargs)
        f = os.path.join(path, "config.json")
        if os.path.isdir(path) and os.path.exists(f):
            with open(f) as fh:
                for k, v in json.load(fh).items():
                    if k in cfg.__dict__ and k not in kwargs:
                        setattr(cfg, k, v)
        return cfg


class MVLBertConfigforVQA(MVLBertConfig):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.attention_probs_dropout_prob = 0.1
        self.hidden_dropout_prob = 0.1


class MVLBertPretrainConfig(MVLBertConfig):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.ITM_task = kwargs.get("ITM_task", False)
        self.max_length = kwargs.get("max_length", 150)
        self.attention_probs_dropout_prob = 0.1
        self.hidden_dropout_prob = 0.1


class MVLBertRetrieval(MVLBertConfig):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.lr = 1e-6
        self.max_length = kwargs.get("max_length", 80)
        self.attention_probs_dropout_prob = 0.1


class MVLBertConfigForImageCaption(MVLBertConfig):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.lr = 1e-5
        self.max_length = kwargs.get("max_length", 80)
        self.is_decoder = True
        self.attention_probs_dropout_prob = 0.1
        self.hidden_dropout_prob = 0.1


# ----------------------------------------------------------------------------- Conv_layer (model.py:186-266)
class _GeluMarker(nn.GELU):
    """nn.GELU placeholder at ``conv.1``; the activation is fused into the
    final Swin LayerNorm kernel."""


class Conv_layer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.hidden_size = config.hidden_size if config is not None else 768
        if config.conv.lower() != 'swintransformer':
            raise NotImplementedError("only config.conv='swintransformer' is on the MI355X hot path")
        conv = SwinTransformer(**config.swin)
        ckpt = getattr(config, "swin_checkpoint", None)
        if ckpt is not None and os.path.exists(ckpt):     # reference: torch.load(...)['model'], strict=False (model.py:222-226)
            conv.load_state_dict(torch.load(ckpt, map_location='cpu')['model'], strict=False)
        self.conv = nn.Sequential(conv, _GeluMarker())
        self.resnet_fc = nn.Linear(2048, config.hidden_size)    # unused with Swin; kept for state-dict parity
        self.feature_proj = None
        if conv.num_features != config.hidden_size:
            if not getattr(config, "swin_feature_proj", False):
                raise ValueError(f"Swin emits {conv.num_features}-d tokens but hidden_size is {config.hidden_size} "
                                 "(the reference has no projection for Swin features either, model.py:263); "
                                 "set config.swin_feature_proj = True / config.use_swin_base() for the added Linear")
            self.feature_proj = nn.Linear(conv.num_features, config.hidden_size)     # build-added (config #5)

    def forward(self, v):
        swin = self.conv[0]
        if torch.is_tensor(v) and v.dim() == 5:       # IU-Xray image pairs (model.py:240-253)
            # both views go through the (shared-weight) Swin as ONE batch of n*B images: one backward pass
            # produces the summed weight gradient, and the kernels see twice the rows
            B, n = v.shape[0], v.shape[1]
            f = self._project(swin(v.transpose(0, 1).reshape(n * B, *v.shape[2:]), fuse_gelu=True))   # [n*B, 49, H]
            return f.view(n, B, f.shape[1], f.shape[2]).transpose(0, 1).reshape(B, n * f.shape[1], f.shape[2])
        return self._project(swin(v, fuse_gelu=True))

    def _project(self, f):
        if self.feature_proj is None:
            return f
        B, n, C = f.shape
        out = _FeatureProjFn.apply(_token(self.feature_proj, f.device), f.reshape(B * n, C).contiguous(),
                                   self.feature_proj, torch.is_grad_enabled())
        return out.view(B, n, -1)


# ----------------------------------------------------------------------------- MLM head holders (HF BertOnlyMLMHead)
class BertPredictionHeadTransform(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)


class BertLMPredictionHead(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.transform = BertPredictionHeadTransform(config)
        self.decoder = nn.Linear(config.hidden_size, config.vocab_size, bias=True)
        self.bias = nn.Parameter(torch.zeros(config.vocab_size))   # present in HF state dicts; decoder.bias is what forward uses


class BertOnlyMLMHead(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.predictions = BertLMPredictionHead(config)

    def forward(self, sequence_output):
        """logits [..., V] (differentiable) -- used by the caption head / decode."""
        shp = sequence_output.shape
        x = sequence_output.reshape(-1, shp[-1]).contiguous()
        tok = _token(self, x.device)
        return _MlmLogitsFn.apply(tok, x, self, torch.is_grad_enabled()).view(*shp[:-1], -1)

    # ---- engine pieces shared by the fused-loss path and the logits path
    # rd (optional): int32 device scalar, the number of valid rows of x (labelled rows gathered first; MvltGemm.m_dev)
    def _transform(self, ar, x, save, rd=None):
        pr = self.predictions
        pre = torch.empty_like(x)
        t1 = ops.gemm(x, ar.compute(pr.transform.dense.weight), bias=pr.transform.dense.bias.data, gelu=True, save_pre=pre,
                      m_dev=rd)
        ln = pr.transform.LayerNorm
        t2, mean, rstd, _ = ops.layernorm_fwd(t1, ln.weight.data, ln.bias.data, ln.eps, save_stats=save, rows_dev=rd)
        return pre, t1, t2, mean, rstd

    def _logits(self, ar, t2, rd=None):
        pr = self.predictions
        V = pr.decoder.out_features
        ld = (V + 63) // 64 * 64
        logits = ops.gemm(t2, ar.compute(pr.decoder.weight), bias=pr.decoder.bias.data, ldc=ld, m_dev=rd)
        return logits, V

    def _backward_from_dlogits(self, ar, dlogits, V, x, pre, t1, t2, mean, rstd, rd=None):
        pr = self.predictions
        g = ar.grad_view
        dl = dlogits[:, :V]
        # with the labelled rows first only a few row tiles are live, each with a 30522-deep reduction: split it
        dt2 = ops.gemm(dl, ar.compute(pr.decoder.weight), b_kmajor=True, m_dev=rd, split_k=8 if rd is not None else 0)
        ops.gemm(dl, t2, a_kmajor=True, b_kmajor=True, out=g(pr.decoder.weight), out_f32=True,
                 a_colsum=g(pr.decoder.bias), m_dev=rd)
        ln = pr.transform.LayerNorm
        dt1 = ops.layernorm_bwd(dt2, t1, mean, rstd, ln.weight.data, g(ln.weight), g(ln.bias), rows_dev=rd)
        dpre = ops.gelu_bwd(pre, dt1)
        # rows beyond the device-side count are not computed: their gradient is exactly zero (unlabelled rows)
        dx = ops.gemm(dpre, ar.compute(pr.transform.dense.weight), b_kmajor=True, m_dev=rd,
                      out=None if rd is None else torch.zeros_like(x))
        ops.gemm(dpre, x, a_kmajor=True, b_kmajor=True, out=g(pr.transform.dense.weight), out_f32=True,
                 a_colsum=g(pr.transform.dense.bias), m_dev=rd)
        ar.mark(pr.decoder.weight, pr.decoder.bias, ln.weight, ln.bias, pr.transform.dense.weight, pr.transform.dense.bias)
        return dx


def _token(mod, device):
    tok = mod.__dict__.get("_mvlt_token")
    if tok is None or tok.device != device:
        tok = torch.zeros(1, device=device, requires_grad=True)
        mod.__dict__["_mvlt_token"] = tok
    return tok


class _MlmLogitsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, token, x, head, save):
        ar = Arena.of(head, x.dtype)
        ar.refresh_shadow()
        pre, t1, t2, mean, rstd = head._transform(ar, x, save)
        logits, V = head._logits(ar, t2)
        ctx.head, ctx.saved = head, (ar, V, x, pre, t1, t2, mean, rstd) if save else None
        return logits[:, :V]

    @staticmethod
    def backward(ctx, dlogits):
        ar, V, x, pre, t1, t2, mean, rstd = ctx.saved
        backward_begin(ar)
        ld = (V + 63) // 64 * 64
        dfull = torch.zeros((dlogits.shape[0], ld), dtype=x.dtype, device=x.device)
        dfull[:, :V] = dlogits
        dx = ctx.head._backward_from_dlogits(ar, dfull, V, x, pre, t1, t2, mean, rstd)
        ctx.saved = None
        return None, dx, None, None


class _MlmLossFn(torch.autograd.Function):
    """MLM head + F.cross_entropy(ignore_index=-100) fused (model.py:399-410):
    logits stay in one padded [rows, ld] buffer, CE backward overwrites it in place."""

    @staticmethod
    def forward(ctx, token, x, head, labels, save, rd=None, label_sync=None):
        ar = Arena.of(head, x.dtype)
        ar.refresh_shadow()
        with ops.pin_stream():
            pre, t1, t2, mean, rstd = head._transform(ar, x, save, rd)
            logits, V = head._logits(ar, t2, rd)
            acc, lse = ops.ce_fwd(logits, V, labels, rows_dev=rd)
        if label_sync is not None:
            # data parallel: model.py:410 is a mean over the labelled tokens of the WHOLE batch.  label_sync all-reduces
            # the 4-byte label count and returns N_global / world; dividing this rank's nll sum by it makes the rank
            # average of the per-rank losses (and of their gradients: GradReducer averages) the global-batch mean.
            acc = torch.cat([acc[:1], label_sync(acc[1:2])])
        loss = acc[0] / acc[1]          # mean over labelled positions (nan if none, as in torch)
        ctx.head = head
        ctx.saved = (ar, V, x, pre, t1, t2, mean, rstd, logits, labels, lse, acc, rd) if save else None
        return loss

    @staticmethod
    def backward(ctx, dloss):
        ar, V, x, pre, t1, t2, mean, rstd, logits, labels, lse, acc, rd = ctx.saved
        backward_begin(ar)
        gs = dloss.reshape(1).to(torch.float32).contiguous()
        with ops.pin_stream():
            dlogits = ops.ce_bwd(logits, V, labels, lse, acc, grad_scale=1.0, grad_scale_dev=gs, rows_dev=rd)
            dx = ctx.head._backward_from_dlogits(ar, dlogits, V, x, pre, t1, t2, mean, rstd, rd)
        ctx.saved = None
        return None, dx, None, None, None, None, None


class _GatherRowsFn(torch.autograd.Function):
    """x = hidden[gather_row] with the backward pass of the labelled-row head: only the first `count` gathered rows
    carry a gradient and their source rows are distinct, so the backward pass is one scatter into a zeroed buffer
    (torch's index_put_(accumulate=True) sorts the indices: a radix sort plus ~50 tiny copies per step)."""

    @staticmethod
    def forward(ctx, hidden, gather_row, count):
        ctx.save_for_backward(gather_row, count)
        ctx.shape = hidden.shape
        with ops.pin_stream():
            return ops.rows_transform(hidden, rowmap=gather_row)

    @staticmethod
    def backward(ctx, dx):
        gather_row, count = ctx.saved_tensors
        dh = torch.zeros(ctx.shape, dtype=dx.dtype, device=dx.device)
        with ops.pin_stream():
            ops.rows_scatter(dx.contiguous(), gather_row, count, dh)
        return dh, None, None


class _LinearCEFn(torch.autograd.Function):
    """Small classifier + cross entropy (ITM, model.py:415-418)."""

    @staticmethod
    def forward(ctx, token, x, lin, labels, save):
        ar = Arena.of(lin, x.dtype)
        ar.refresh_shadow()
        V = lin.out_features
        ld = (V + 3) // 4 * 4
        logits = ops.gemm(x, ar.compute(lin.weight), bias=lin.bias.data, ldc=ld)
        acc, lse = ops.ce_fwd(logits, V, labels)
        ctx.lin, ctx.saved = lin, (ar, V, x, logits, labels, lse, acc) if save else None
        return acc[0] / acc[1]

    @staticmethod
    def backward(ctx, dloss):
        ar, V, x, logits, labels, lse, acc = ctx.saved
        backward_begin(ar)
        lin = ctx.lin
        gs = dloss.reshape(1).to(torch.float32).contiguous()
        dl = ops.ce_bwd(logits, V, labels, lse, acc, grad_scale_dev=gs)[:, :V]
        dx = ops.gemm(dl, ar.compute(lin.weight), b_kmajor=True)
        ops.gemm(dl, x, a_kmajor=True, b_kmajor=True, out=ar.grad_view(lin.weight), out_f32=True,
                 a_colsum=ar.grad_view(lin.bias))
        ar.mark(lin.weight, lin.bias)
        ctx.saved = None
        return None, dx, None, None, None


class _FeatureProjFn(torch.autograd.Function):
    """Linear in the compute dtype (Swin-B 1024-d tokens -> hidden_size)."""

    @staticmethod
    def forward(ctx, token, x, lin, save):
        ar = Arena.of(lin, x.dtype)
        ar.refresh_shadow()
        ctx.lin, ctx.saved = lin, (ar, x) if save else None
        return ops.gemm(x, ar.compute(lin.weight), bias=lin.bias.data)

    @staticmethod
    def backward(ctx, dy):
        ar, x = ctx.saved
        backward_begin(ar)
        lin = ctx.lin
        dy = dy.contiguous()
        dx = ops.gemm(dy, ar.compute(lin.weight), b_kmajor=True)
        ops.gemm(dy, x, a_kmajor=True, b_kmajor=True, out=ar.grad_view(lin.weight), out_f32=True,
                 a_colsum=ar.grad_view(lin.bias))
        ar.mark(lin.weight, lin.bias)
        ctx.saved = None
        return None, dx, None, None


class _LinearFn(torch.autograd.Function):
    """(dropout ->) Linear producing f32 logits (VQA / retrieval final layer)."""

    @staticmethod
    def forward(ctx, token, x, lin, p_drop, save):
        ar = Arena.of(lin, x.dtype)
        ar.refresh_shadow()
        seed = next_seed() if p_drop > 0 else 0
        xd = ops.rows_transform(x, dropout=(p_drop, seed, 4001)) if p_drop > 0 else x
        V = lin.out_features
        logits = ops.gemm(xd, ar.compute(lin.weight), bias=lin.bias.data, ldc=(V + 3) // 4 * 4)
        ctx.lin, ctx.saved = lin, (ar, V, xd, p_drop, seed) if save else None
        return ops.cast(logits, torch.float32)[:, :V] if logits.dtype != torch.float32 else logits[:, :V]

    @staticmethod
    def backward(ctx, dlogits):
        ar, V, xd, p_drop, seed = ctx.saved
        backward_begin(ar)
        lin = ctx.lin
        dl = dlogits.to(xd.dtype).contiguous()
        dxd = ops.gemm(dl, ar.compute(lin.weight), b_kmajor=True)
        ops.gemm(dl, xd, a_kmajor=True, b_kmajor=True, out=ar.grad_view(lin.weight), out_f32=True,
                 a_colsum=ar.grad_view(lin.bias))
        ar.mark(lin.weight, lin.bias)
        dx = ops.rows_transform(dxd, dropout=(p_drop, seed, 4001)) if p_drop > 0 else dxd
        ctx.saved = None
        return None, dx, None, None, None


# ----------------------------------------------------------------------------- base with HF-like persistence
class MVLBertPretrainedModel(nn.Module):
    base_model_prefix = "MVLBert"
    config_class = MVLBertConfig

    def __init__(self, config):
        super().__init__()
        self.config = config

    @staticmethod
    def check_device_errors():
        """Raise if an in-launch hand-off of the fused Swin attention (mvlt_swin_wmsa2_fwd) ever timed out on this process'
        devices (ops.DeviceHandoffError; the affected activations are NaN).  Synchronises the device: call it where the caller
        synchronises anyway (after loss.item(), before writing a checkpoint)."""
        ops.wmsa2_check(sync=True)

    def _flush_optimizer_tail(self):
        """A deferred optimizer tail (optim.FusedAdamW(defer_tail=True)) is applied before parameters are read or replaced."""
        ar = self.__dict__.get("_mvlt_arena")
        if ar is not None and ar.__dict__.get("_opt_tail") is not None:
            ar._opt_tail.flush()

    def state_dict(self, *args, **kwargs):
        self._flush_optimizer_tail()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._flush_optimizer_tail()
        return super().load_state_dict(*args, **kwargs)

    def save_pretrained(self, path):
        if any(p.is_cuda for p in self.parameters()):
            self.check_device_errors()          # (the copy to the host below synchronises anyway) never persist poisoned weights
        os.makedirs(path, exist_ok=True)
        torch.save({k: v.detach().cpu() for k, v in self.state_dict().items()}, os.path.join(path, "pytorch_model.bin"))
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(self.config.to_dict(), f)

    @classmethod
    def from_pretrained(cls, path, config=None, **kwargs):
        config = config or cls.config_class.from_pretrained(path)
        model = cls(config, **kwargs)
        f = os.path.join(path, "pytorch_model.bin")
        if os.path.exists(f):
            sd = torch.load(f, map_location="cpu")
            for head in ("MLM_head_seq2seq", "MLM_head_bidir"):     # HF 4.x ties predictions.bias == decoder.bias
                b, d = f"{head}.predictions.bias", f"{head}.predictions.decoder.bias"
                if b in sd and d not in sd:
                    sd[d] = sd[b].clone()
            model.load_state_dict(sd, strict=False)
        return model


# ----------------------------------------------------------------------------- heads
_AUTO_PACK = os.environ.get("MVLT_AUTO_PACK", "1") != "0"




class MVLBertForPretraining(MVLBertPretrainedModel):
    """model.py:352-420.  The seq2seq/bidirectional coin flip (model.py:390-394)
    uses Python's ``random`` like the reference; under DDP every rank must draw
    the same value (mvlt_amd.ddp seeds it)."""

    def __init__(self, config):
        super().__init__(config)
        self.config.output_text_and_image_seperately = True
        self.conv = Conv_layer(config)
        self.MVLBert = MVLBert(config, add_pooling_layer=True)
        self.MLM_head_seq2seq = BertOnlyMLMHead(config)
        self.MLM_head_bidir = BertOnlyMLMHead(config)
        self.ITM_mlp = nn.Linear(config.hidden_size, 2)
        self.last_seq2seq = None

    def forward(self, image, caption_masked, caption_label, image_text_label, image_mask=None, text_lengths=None):
        """Reference signature (model.py:372) plus one optional argument: ``text_lengths`` (host ints [B], the
        tokeniser's caption lengths).  When given, the encoder runs on packed rows -- the zero-padded tail of
        every caption is not materialised (MVLBert.forward_packed); loss and gradients are those of the dense
        computation.  A non-zero id or a label at or beyond the stated length turns the loss into NaN."""
        Arena.of(self, compute_dtype_of(self))       # one arena for the whole model
        # the packing / label plans depend on the ids and labels only: MVLBert.forward_autopack runs their two
        # single-workgroup kernels on the side stream behind this point, i.e. beside the image tower instead of after it
        entry = None
        if caption_masked.is_cuda:
            entry = torch.cuda.Event()
            entry.record()
        image_feature = self.conv(image)
        text_idx = caption_masked
        text_mask = None          # == (text_idx > 0); rebuilt in-kernel from the ids
        seq2seq_mask = random.random() < 0.5
        self.last_seq2seq = seq2seq_mask
        packed = text_lengths is not None and image_mask is None
        # default: the packing plan is derived ON THE DEVICE from the ids and labels themselves (no new argument, no
        # host sync; config.auto_pack_rows = False or MVLT_AUTO_PACK=0 computes every padded row like the reference)
        auto = (not packed and image_mask is None and getattr(self.config, "auto_pack_rows", True) and _AUTO_PACK
                and text_idx.dtype == torch.int64)
        B, T = text_idx.shape
        n_img = image_feature.shape[1]
        text_row = None
        if auto:
            lab_first_early = (self.config.MLM_task and getattr(self.config, "mlm_labelled_rows_first", True)
                               and not (getattr(self.config, "mlm_max_labels_per_sample", None) is not None
                                        and self.config.mlm_max_labels_per_sample * B < B * T))
            hidden, pooled, text_row = self.MVLBert.forward_autopack(
                text_idx, image_feature, labels=caption_label if self.config.MLM_task else None, seq2seq_mask=seq2seq_mask,
                inputs_ready=entry, want_label_plan=lab_first_early)
            H = hidden.shape[1]
        elif packed:
            hidden, pooled, row_start, seq_len = self.MVLBert.forward_packed(text_idx, image_feature, text_lengths,
                                                                             seq2seq_mask=seq2seq_mask)
            H = hidden.shape[1]
            beyond = torch.arange(T, device=hidden.device)[None, :] >= (seq_len.to(torch.int64) - (n_img + 2))[:, None]
            bad_len = (((text_idx != 0) | (caption_label.reshape(B, T) >= 0)) & beyond).any()
        else:
            text_out, _, pooled, _ = self.MVLBert(text_idx, text_mask, image_feature, image_mask,
                                                  seq2seq_mask=seq2seq_mask, output_text_image_seperate=True)
            H = text_out.shape[2]
        head = self.MLM_head_seq2seq if seq2seq_mask else self.MLM_head_bidir
        dev = image_feature.device
        mlm_loss = torch.zeros((1, 1))
        itm_loss = None
        if self.config.MLM_task:
            labels = caption_label.reshape(-1).to(torch.int64).contiguous()
            cap = getattr(self.config, "mlm_max_labels_per_sample", None)
            compact = cap is not None and cap * B < B * T
            # default: labelled rows are gathered FIRST and their number stays on the device; the head's kernels read
            # it (MvltGemm.m_dev) -- loss and gradients are those of the all-rows head (ignore_index rows contribute
            # nothing, model.py:410), for any number of labels, with no host sync and no capacity to configure
            rd = None
            lab_first = auto and not compact and getattr(self.config, "mlm_labelled_rows_first", True)
            if lab_first:
                early = self.MVLBert.__dict__.pop("_mvlt_label_plan", None)
                if early is not None:
                    gather_row, sel_labels, rd = early
                else:
                    with ops.pin_stream():
                        gather_row, sel_labels, rd = ops.label_plan(labels, text_row)
            if compact:
                # Only labelled positions contribute to F.cross_entropy(ignore_index=-100) (model.py:410),
                # so the MLM head (768x30522 decoder, 312 MB of f32 logits in the reference) is evaluated on
                # a fixed-capacity gather of them: a stable sort puts labelled rows first, the padding rows
                # carry label -100 and are ignored.  No host sync; more than `cap` labels per sample on
                # average would drop rows, which turns the loss into NaN instead of a silently wrong value.
                valid = labels >= 0
                order = torch.argsort((~valid).to(torch.int8), stable=True)[: cap * B]
                sel_labels = labels[order].contiguous()
            if lab_first:
                x = _GatherRowsFn.apply(hidden, gather_row, rd)
            elif auto:
                # packed row of every caption position (dropped positions: the sample's [CLS] row; label -100)
                x = hidden[text_row[order] if compact else text_row]
            elif packed:
                # flat (b, t) -> packed row of caption position t; rows of unlabelled picks are clamped into
                # range (their label is -100, whatever they gather is ignored)
                flat = order if compact else torch.arange(B * T, device=dev)
                b_of = torch.div(flat, T, rounding_mode="floor")
                rows = (row_start[b_of] + (n_img + 2) + (flat - b_of * T)).clamp_(max=hidden.shape[0] - 1)
                x = hidden[rows]
            elif compact:
                x = text_out.reshape(B * T, H)[order]
            else:
                x = text_out.reshape(B * T, H)
            mlm_loss = _MlmLossFn.apply(_token(head, dev), x.contiguous(), head,
                                        sel_labels if (compact or lab_first) else labels, torch.is_grad_enabled(), rd,
                                        # the label-count all-reduce is a COLLECTIVE: only a forward that records a graph (one
                                        # every rank runs and follows with backward()) takes it; no_grad forwards -- rank-0
                                        # validation, an uneven last evaluation batch -- keep the per-rank mean and never touch
                                        # the process group (GradReducer.no_label_sync() switches it off for grad-mode forwards too)
                                        self.__dict__.get("_mvlt_label_sync") if torch.is_grad_enabled() else None)
            if compact:
                mlm_loss = torch.where(valid.sum() > cap * B, torch.full_like(mlm_loss, float("nan")), mlm_loss)
            if packed:
                mlm_loss = torch.where(bad_len, torch.full_like(mlm_loss, float("nan")), mlm_loss)
        if self.config.ITM_task:
            itm_loss = _LinearCEFn.apply(_token(self.ITM_mlp, dev), pooled, self.ITM_mlp,
                                         image_text_label.reshape(-1).to(torch.int64).contiguous(),
                                         torch.is_grad_enabled())
        if itm_loss is None:
            return mlm_loss
        # model.py:419 takes .mean() of both: they are scalars already (a mean over ONE element is a reduction launch forward
        # and a scaling launch backward each)
        return (mlm_loss if mlm_loss.dim() == 0 else mlm_loss.mean()) + (itm_loss if itm_loss.dim() == 0 else itm_loss.mean())


class MVLBertForVQA(MVLBertPretrainedModel):
    """model.py:297-349 -> (prob, logits)."""

    def __init__(self, config):
        super().__init__(config)
        self.conv = Conv_layer(config)
        self.activation = nn.GELU()
        self.add_pooling_layer = True
        self.MVLBert = MVLBert(config, add_pooling_layer=True)
        self.final_mlp = nn.Sequential(nn.Dropout(config.hidden_dropout_prob, inplace=False),
                                       nn.Linear(config.hidden_size, config.result_num))
        self.softmax = nn.Softmax(dim=-1)

    def forward(self, image, question, label, image_mask=None):
        """``config.eval_cuda_graph = True`` (opt-in): in eval mode without autograd the forward is captured once per input
        shape as a HIP graph and replayed (runtime.GraphedEval) -- the inference call of run_vqa.py at B = 2 is host-enqueue
        bound otherwise (~330 launches).  Same kernels, bit-identical results."""
        if (getattr(self.config, "eval_cuda_graph", False) and not self.training and not torch.is_grad_enabled()
                and image.is_cuda and question is not None):
            ge = self.__dict__.get("_mvlt_graphed")
            if ge is None:
                ge = self.__dict__["_mvlt_graphed"] = GraphedEval(lambda im, q, mk: self._forward_impl(im, q, mk), self)
            return ge(image, question, image_mask)
        return self._forward_impl(image, question, image_mask)

    def _forward_impl(self, image, question, image_mask=None):
        Arena.of(self, compute_dtype_of(self))
        image_feature = self.conv(image)
        _, pooled = self.MVLBert(text_idx=question, text_mask=None, image_feature=image_feature, image_mask=image_mask)
        lin = self.final_mlp[1]
        p = self.final_mlp[0].p if self.training else 0.0
        logits = _LinearFn.apply(_token(lin, pooled.device), pooled, lin, p, torch.is_grad_enabled())
        with torch.no_grad():
            prob = ops.softmax_rows(logits.contiguous(), logits.shape[1])
        return prob, logits


class MVLBertForRetrieval(MVLBertPretrainedModel):
    """model.py:423-476: transform (dense+GELU+LN) + Linear(768,2)."""

    def __init__(self, config):
        super().__init__(config)
        self.conv = Conv_layer(config)
        self.MVLBert = MVLBert(config, add_pooling_layer=True)
        self.final_mlp = nn.Sequential(BertPredictionHeadTransform(config), nn.Linear(config.hidden_size, 2))

    def forward(self, image, caption, image_text_label=None, image_mask=None):
        """model.py:444-476: logits = Linear(LN(GELU(dense(pooled)))); softmax prob when no label is given.
        (The retrieval *drivers* are out of scope; the head reuses the hot path.)"""
        Arena.of(self, compute_dtype_of(self))
        image_feature = self.conv(image)
        _, pooled = self.MVLBert(text_idx=caption, text_mask=None, image_feature=image_feature, image_mask=image_mask)
        tr, lin = self.final_mlp[0], self.final_mlp[1]
        logits = _TransformLinearFn.apply(_token(lin, pooled.device), pooled, tr, lin, torch.is_grad_enabled())
        if image_text_label is None:
            with torch.no_grad():
                return ops.softmax_rows(logits.contiguous(), logits.shape[1])
        return logits


class _TransformLinearFn(torch.autograd.Function):
    """BertPredictionHeadTransform (dense + GELU + LayerNorm) followed by a small Linear, f32 logits."""

    @staticmethod
    def forward(ctx, token, x, tr, lin, save):
        ar = Arena.of(lin, x.dtype)
        ar.refresh_shadow()
        x = x.contiguous()
        pre = torch.empty_like(x)
        t1 = ops.gemm(x, ar.compute(tr.dense.weight), bias=tr.dense.bias.data, gelu=True, save_pre=pre)
        t2, mean, rstd, _ = ops.layernorm_fwd(t1, tr.LayerNorm.weight.data, tr.LayerNorm.bias.data, tr.LayerNorm.eps,
                                              save_stats=save)
        V = lin.out_features
        logits = ops.gemm(t2, ar.compute(lin.weight), bias=lin.bias.data, ldc=(V + 3) // 4 * 4)
        ctx.mods, ctx.saved = (tr, lin), (ar, V, x, pre, t1, t2, mean, rstd) if save else None
        return ops.cast(logits, torch.float32)[:, :V] if logits.dtype != torch.float32 else logits[:, :V]

    @staticmethod
    def backward(ctx, dlogits):
        ar, V, x, pre, t1, t2, mean, rstd = ctx.saved
        tr, lin = ctx.mods
        backward_begin(ar)
        g = ar.grad_view
        dl = dlogits.to(x.dtype).contiguous()
        dt2 = ops.gemm(dl, ar.compute(lin.weight), b_kmajor=True)
        ops.gemm(dl, t2, a_kmajor=True, b_kmajor=True, out=g(lin.weight), out_f32=True, a_colsum=g(lin.bias))
        dt1 = ops.layernorm_bwd(dt2, t1, mean, rstd, tr.LayerNorm.weight.data, g(tr.LayerNorm.weight), g(tr.LayerNorm.bias))
        dpre = ops.gelu_bwd(pre, dt1)
        dx = ops.gemm(dpre, ar.compute(tr.dense.weight), b_kmajor=True)
        ops.gemm(dpre, x, a_kmajor=True, b_kmajor=True, out=g(tr.dense.weight), out_f32=True, a_colsum=g(tr.dense.bias))
        ar.mark(lin.weight, lin.bias, tr.LayerNorm.weight, tr.LayerNorm.bias, tr.dense.weight, tr.dense.bias)
        ctx.saved = None
        return None, dx, None, None, None


class MVLBertForImageCaption(MVLBertPretrainedModel):
    """model.py:479-546 (+ greedy_search :826-984 and beam_search :636-816 in decode.py)."""

    def __init__(self, config, tokenizer=None):
        super().__init__(config)
        assert config.is_decoder, 'config.is_decoder should be True if you want to run image caption for testing'
        self.MVLBert = MVLBert(config, add_pooling_layer=True)
        self.conv = Conv_layer(config)
        self.tokenizer = tokenizer
        self.MLM_head_seq2seq = BertOnlyMLMHead(config)

    def forward(self, image, caption, num_beams, learning_strategy, sample_mode='greedy'):
        Arena.of(self, compute_dtype_of(self))
        image_feature = self.conv(image)
        if num_beams > 1:
            from .decode import beam_search          # scorer restated from HF 4.16: parity unpinned
            return beam_search(self, image_feature, num_beams, learning_strategy=learning_strategy)
        if num_beams == 1:
            from .decode import greedy_search
            return greedy_search(self, image_feature, learning_strategy=learning_strategy, sample_mode=sample_mode)
        return self.encode_forward(image_feature, caption, learning_strategy)

    def encode_forward(self, image_feature, caption, learning_strategy):
        text_out, _, _, sep_out = self.MVLBert(text_idx=caption, text_mask=None, image_feature=image_feature,
                                               image_mask=None, seq2seq_mask=True, output_text_image_seperate=True)
        if learning_strategy == 'unilm':
            return self.MLM_head_seq2seq(text_out).transpose(1, 2)
        if learning_strategy == 'normal':
            return self.MLM_head_seq2seq(torch.cat([sep_out[:, None], text_out[:, :-1]], dim=1)).transpose(1, 2)
        raise NotImplementedError("learning_strategy:", learning_strategy, "is not implemented! Try 'unilm' or 'normal'.")
