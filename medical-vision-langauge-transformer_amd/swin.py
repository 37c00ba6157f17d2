"""Swin Transformer backbone, drop-in for the reference's
``modules/visual_feature_extractor.py:125-701`` (same constructor / forward
signatures, same state-dict keys), executed by the gfx950 kernels.

The nn.Module tree below only *holds parameters* (reference names/shapes).
The arithmetic is a hand-scheduled kernel sequence (``_forward`` /
``_backward``) -- 7 launches per block forward:

    LN1 (+ cyclic shift + window partition fused into the store)
    QKV GEMM (+bias)            | fused window attention (bias table, shift mask in-kernel)
    proj GEMM (+bias, DropPath scale, window reverse + un-shift scatter, residual)
    LN2 | fc1 GEMM (+bias, GELU, pre-activation saved) | fc2 GEMM (+bias, DropPath, residual)

The whole backbone is ONE autograd node; its backward walks the blocks in
reverse and writes parameter gradients straight into the arena.
"""
from __future__ import annotations

import math
import os
from typing import List, Optional

import torch
import torch.nn as nn

from . import ops
from . import _lib as L
from .arena import Arena
from .indexing import batched_window_maps, relative_position_index, shift_attn_mask
from .runtime import backward_begin, compute_dtype_of, next_seed


# ----------------------------------------------------------------------------- parameter holders
class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features, in_features)


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, qk_scale=None):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        ws = window_size[0]
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) * (2 * ws - 1), num_heads))
        self.register_buffer("relative_position_index", relative_position_index(ws))
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, input_resolution, num_heads, window_size=7, shift_size=0, mlp_ratio=4.,
                 qk_scale=None, drop_path=0.):
        super().__init__()
        self.dim, self.input_resolution, self.num_heads = dim, tuple(input_resolution), num_heads
        self.window_size, self.shift_size, self.mlp_ratio = window_size, shift_size, mlp_ratio
        if min(self.input_resolution) <= self.window_size:      # reference :302-305
            self.shift_size = 0
            self.window_size = min(self.input_resolution)
        assert 0 <= self.shift_size < self.window_size, "shift_size must in 0-window_size"
        self.drop_path_prob = float(drop_path)
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, (self.window_size, self.window_size), num_heads, qk_scale)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        H, W = self.input_resolution
        mask = shift_attn_mask(H, W, self.window_size, self.shift_size) if self.shift_size > 0 else None
        self.register_buffer("attn_mask", mask)


class PatchMerging(nn.Module):
    def __init__(self, input_resolution, dim):
        super().__init__()
        self.input_resolution, self.dim = tuple(input_resolution), dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)


class BasicLayer(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio, qk_scale, drop_path,
                 downsample):
        super().__init__()
        self.dim, self.input_resolution, self.depth = dim, tuple(input_resolution), depth
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, input_resolution, num_heads, window_size,
                                 0 if i % 2 == 0 else window_size // 2, mlp_ratio, qk_scale,
                                 drop_path[i] if isinstance(drop_path, list) else drop_path)
            for i in range(depth)])
        self.downsample = PatchMerging(input_resolution, dim) if downsample else None


class PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        self.img_size, self.patch_size = (img_size, img_size), (patch_size, patch_size)
        self.patches_resolution = [img_size // patch_size, img_size // patch_size]
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        self.in_chans, self.embed_dim = in_chans, embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.LayerNorm(embed_dim)


_FUSED_WMSA_BWD = False      # mvlt_swin_wmsa_bwd (one-launch backward of the first design) is parity-tested but 1.3-3.5x slower than the
                             # three launches at B = 32: the tests switch it on through this attribute (Python host path)
_FUSED_WMSA = os.environ.get("MVLT_FUSED_WMSA", "auto")     # "0" never, "1" wherever supported, "auto" where it wins
_SWIN_BWD_ONE = int(os.environ.get("MVLT_SWIN_BWD_ONE", "384"))        # smallest width whose blocks take the one-launch backward (0: none; host.cpp reads it too)
_SWIN_BWD_ONE = (1 << 30) if _SWIN_BWD_ONE <= 0 else (96 if _SWIN_BWD_ONE == 1 else _SWIN_BWD_ONE)
_SWIN_BWD_PROJ = os.environ.get("MVLT_SWIN_BWD_PROJ", "1") != "0"     # proj dgrad inside the attention backward (A/B switch; host.cpp reads it too)
_WMSA2 = os.environ.get("MVLT_WMSA2", "1") != "0"           # second design at stage 2 (A/B measurements and parity tests turn it off)


def _wmsa_mode(dtype, B, res, C, nH):
    """How the attention half of a block runs: 0 = four launches, 1 = mvlt_swin_wmsa_fwd (one window per workgroup), 2 =
    mvlt_swin_wmsa2_fwd (two windows x a head group per workgroup, the groups meet inside the launch).  Measured on MI355X
    at B = 32, training mode (scripts/bench_wmsa.py, profiles/r4_wmsa2_*): stage 0 / 1 / 2 = 57.6 / 36.4 / 27.9 us with the
    second design, 60.1 / 46.1 / 60.1 us with the first, ~94 / 62 / 52 us as four launches.  Small launches (fewer than 64
    windows: the B = 2 configurations) keep the first design where it has enough windows, else the four launches.
    MVLT_FUSED_WMSA=0 forces the four launches, MVLT_WMSA2=0 the first design (A/B measurements, parity tests)."""
    if _FUSED_WMSA == "0":
        return 0
    nwin = B * (res // 7) ** 2
    if _WMSA2 and nwin >= 64 and ops.swin_wmsa2_supported(dtype, B, res, C, nH):
        return 2
    if not ops.swin_wmsa_supported(dtype, C, nH):
        return 0
    return 1 if (_FUSED_WMSA == "1" or nwin >= 512 or (nwin >= 256 and C <= 256)) else 0


class _SwinFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, token, img, mod, fuse_gelu, save):
        with ops.pin_stream():
            out, saved = mod._forward(img, fuse_gelu, save)
        ctx.mod, ctx.saved = mod, saved
        return out

    @staticmethod
    def backward(ctx, dout):
        with ops.pin_stream():
            ctx.mod._backward(ctx.saved, dout.contiguous())
        ctx.saved = None
        return None, None, None, None, None


class SwinTransformer(nn.Module):
    """Signature of reference ``SwinTransformer.__init__`` (visual_feature_extractor.py:601-606);
    ``forward(x[B,3,S,S]) -> [B, L_last, C_last]`` un-pooled tokens (:690-693)."""

    def __init__(self, img_size=224, patch_size=4, in_chans=3, num_classes=1000, embed_dim=96,
                 depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=7, mlp_ratio=4., qkv_bias=True,
                 qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.1, norm_layer=nn.LayerNorm,
                 ape=False, patch_norm=True, use_checkpoint=False, **kwargs):
        super().__init__()
        if ape or not patch_norm or not qkv_bias or norm_layer is not nn.LayerNorm:
            raise NotImplementedError("mvlt_amd Swin supports the reference hot-path options only "
                                      "(ape=False, patch_norm=True, qkv_bias=True, LayerNorm)")
        if drop_rate != 0. or attn_drop_rate != 0.:
            raise NotImplementedError("reference runs Swin with drop_rate = attn_drop_rate = 0")
        if window_size != 7 or any(embed_dim * 2 ** i // h != 32 for i, h in enumerate(num_heads)):
            raise NotImplementedError("window attention kernel is built for window 7, head_dim 32")
        self.num_classes, self.num_layers, self.embed_dim = num_classes, len(depths), embed_dim
        self.ape, self.patch_norm, self.mlp_ratio = ape, patch_norm, mlp_ratio
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.patches_resolution = self.patch_embed.patches_resolution
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            res = (self.patches_resolution[0] // 2 ** i, self.patches_resolution[1] // 2 ** i)
            self.layers.append(BasicLayer(int(embed_dim * 2 ** i), res, depths[i], num_heads[i], window_size,
                                          mlp_ratio, qk_scale, dpr[sum(depths[:i]):sum(depths[:i + 1])],
                                          downsample=i < self.num_layers - 1))
        self.norm = nn.LayerNorm(self.num_features)
        self.avgpool = nn.AdaptiveAvgPool1d(1)
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
        self.apply(self._init_weights)
        self.last_droppath = None        # [n_blocks*2, B] keep/(1-p) scales of the last training forward

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'absolute_pos_embed'}

    @torch.jit.ignore
    def no_weight_decay_keywords(self):
        return {'relative_position_bias_table'}

    def flops(self):
        """MACs of one image, same accounting as the reference (:695-701)."""
        H, W = self.patches_resolution
        f = H * W * self.embed_dim * self.patch_embed.in_chans * 16 + H * W * self.embed_dim
        for layer in self.layers:
            h, w = layer.input_resolution
            d = layer.dim
            for blk in layer.blocks:
                n = blk.window_size ** 2
                per_win = n * d * 3 * d + 2 * blk.num_heads * n * n * (d // blk.num_heads) + n * d * d
                f += 2 * d * h * w + (h * w / n) * per_win + 2 * h * w * d * d * blk.mlp_ratio
            if layer.downsample is not None:
                f += h * w * d + (h // 2) * (w // 2) * 4 * d * 2 * d
        f += self.num_features * H * W // (2 ** self.num_layers) + self.num_features * self.num_classes
        return f

    # ------------------------------------------------------------------ public forward
    def forward_features(self, x):
        return self.forward(x)

    def forward(self, x, fuse_gelu: bool = False):
        B, C, H, W = x.shape
        assert H == self.patch_embed.img_size[0] and W == self.patch_embed.img_size[1], \
            f"Input image size ({H}*{W}) doesn't match model ({self.patch_embed.img_size[0]}*{self.patch_embed.img_size[1]})."
        if not x.is_cuda:
            raise RuntimeError("mvlt_amd runs on the GPU only (no CPU fallback)")
        tok = self.__dict__.get("_mvlt_token")
        if tok is None or tok.device != x.device:
            tok = torch.zeros(1, device=x.device, requires_grad=True)
            self.__dict__["_mvlt_token"] = tok
        return _SwinFn.apply(tok, x.contiguous().float(), self, fuse_gelu, torch.is_grad_enabled())

    # ------------------------------------------------------------------ engine
    def _blocks(self):
        for layer in self.layers:
            for blk in layer.blocks:
                yield layer, blk

    def _forward(self, img, fuse_gelu, save):
        cd = compute_dtype_of(self)
        ar = Arena.of(self, cd)
        ar.refresh_shadow(tail=False)          # (Swin parameters are never part of a deferred optimizer tail)
        B = img.shape[0]
        pe = self.patch_embed
        P = pe.patch_size[0]
        train = self.training
        saved = {"B": B, "blocks": [], "merges": [], "train": train} if save else None
        # ---- patch embed: im2col + GEMM(+bias) + LN  (reference :557-565)
        cols = ops.im2col_patch(img, cd, P)
        x0 = ops.gemm(cols, ar.compute(pe.proj.weight), bias=pe.proj.bias.data)
        x, mean, rstd, _ = ops.layernorm_fwd(x0, pe.norm.weight.data, pe.norm.bias.data, pe.norm.eps, save_stats=save)
        if save:
            saved["pe"] = (cols, x0, mean, rstd)
        nblk = sum(len(l.blocks) for l in self.layers)
        dp = None
        if train and any(b.drop_path_prob > 0 for _, b in self._blocks()):
            probs = self.__dict__.get("_dp_probs")
            if probs is None or probs.device != img.device:
                probs = torch.tensor([b.drop_path_prob for _, b in self._blocks() for _ in (0, 1)], device=img.device)
                self.__dict__["_dp_probs"] = probs          # cached: a per-step H2D copy would stall the host
            if img.is_cuda:      # one launch (counter RNG, like the dropout masks) instead of rand / compare / cast / sub / div
                dp = ops.droppath_scales(probs, B, next_seed(), 0x44500000)
            else:
                keep = (torch.rand(2 * nblk, B, device=img.device) >= probs[:, None]).float()
                dp = (keep / (1.0 - probs[:, None])).contiguous()
            self.last_droppath = dp
        bi = 0
        for layer in self.layers:
            H, W = layer.input_resolution
            Lt = H * W
            for blk in layer.blocks:
                s1 = dp[2 * bi] if (dp is not None and blk.drop_path_prob > 0) else None
                s2 = dp[2 * bi + 1] if (dp is not None and blk.drop_path_prob > 0) else None
                x, sv = self._block_fwd(ar, blk, x, B, H, W, s1, s2, save)
                if save:
                    saved["blocks"].append(sv)
                bi += 1
            if layer.downsample is not None:
                ds = layer.downsample
                xm, mean, rstd, _ = ops.layernorm_fwd(x.view(B, Lt, layer.dim), ds.norm.weight.data, ds.norm.bias.data,
                                                      ds.norm.eps, merge=(H, W), save_stats=save)
                xm = xm.view(B * Lt // 4, 4 * layer.dim)
                y = ops.gemm(xm, ar.compute(ds.reduction.weight))
                if save:
                    saved["merges"].append((x, xm, mean, rstd))
                x = y
        y, mean, rstd, ypre = ops.layernorm_fwd(x, self.norm.weight.data, self.norm.bias.data, self.norm.eps,
                                                gelu=fuse_gelu, save_pre=fuse_gelu and save, save_stats=save)
        if save:
            saved["final"] = (x, mean, rstd, ypre)
            saved["arena"] = ar
        return y.view(B, -1, self.num_features), saved

    def _block_fwd(self, ar, blk, x, B, H, W, s1, s2, save):
        C, nH, ws = blk.dim, blk.num_heads, blk.window_size
        Lt = H * W
        nW = (H // ws) * (W // ws)
        w2n, n2w = batched_window_maps(B, H, W, ws, blk.shift_size, x.device)
        at = blk.attn
        if ops.NATIVE:
            # one native call per block (csrc/host.cpp swin_block_fwd)
            w, f, _ = self._block_desc(ar, blk)
            mode = _wmsa_mode(x.dtype, B, H, C, nH)
            geo = [B, H, C, nH, blk.shift_size, mode,
                   ops.wmsa2_sync_ws(x.device, L.lib().mvlt_swin_wmsa2_sync_words(B, H)).data_ptr() if mode == 2 else 0]
            out = ops.host().swin_block_fwd(x, w, f, geo, [w2n.data_ptr(), n2w.data_ptr()], at.scale, blk.norm1.eps,
                                            0 if s1 is None else s1.data_ptr(), 0 if s2 is None else s2.data_ptr(),
                                            save, ops.stream_int())
            return out[0], ((blk, out[1:], s1, s2, H, W) if save else None)
        mode = _wmsa_mode(x.dtype, B, H, C, nH)
        if mode:
            # norm1 + shift/partition + qkv + window attention + proj + reverse + DropPath + residual: one launch
            x1, fs = (ops.swin_wmsa2_fwd if mode == 2 else ops.swin_wmsa_fwd)(x, w2n, B, H, nH, blk.shift_size, blk.norm1.weight.data, blk.norm1.bias.data,
                                       blk.norm1.eps, ar.compute(at.qkv.weight), at.qkv.bias.data,
                                       ar.compute(at.proj.weight), at.proj.bias.data,
                                       at.relative_position_bias_table.data, at.scale, rowscale=s1, save=save)
            xn1w, qkv, ao, lse, mean1, rstd1 = fs if save else (None,) * 6
        else:
            xn1w, mean1, rstd1, _ = ops.layernorm_fwd(x, blk.norm1.weight.data, blk.norm1.bias.data, blk.norm1.eps,
                                                      out_rowmap=n2w, save_stats=save)
            qkv = ops.gemm(xn1w, ar.compute(at.qkv.weight), bias=at.qkv.bias.data)
            ao, lse = ops.attn_fwd(qkv, L.ATTN_SWIN, B * nW, ws * ws, nH, C // nH, at.scale,
                                   bias_table=at.relative_position_bias_table.data, nW=nW, win_res=H,
                                   shift=blk.shift_size)
            x1 = ops.gemm(ao, ar.compute(at.proj.weight), bias=at.proj.bias.data, residual=x, rowmap=w2n,
                          rowscale=(s1, Lt) if s1 is not None else None)
        xn2, mean2, rstd2, _ = ops.layernorm_fwd(x1, blk.norm2.weight.data, blk.norm2.bias.data, blk.norm2.eps,
                                                 save_stats=save)
        h = torch.empty((B * Lt, blk.mlp.fc1.out_features), dtype=x.dtype, device=x.device)
        a = ops.gemm(xn2, ar.compute(blk.mlp.fc1.weight), bias=blk.mlp.fc1.bias.data, gelu=True, save_pre=h)
        x2 = ops.gemm(a, ar.compute(blk.mlp.fc2.weight), bias=blk.mlp.fc2.bias.data, residual=x1,
                      rowscale=(s2, Lt) if s2 is not None else None)
        sv = (blk, x, mean1, rstd1, xn1w, qkv, ao, lse, x1, mean2, rstd2, xn2, h, a, s1, s2, H, W) if save else None
        return x2, sv

    def _backward(self, saved, dout):
        ar: Arena = saved["arena"]
        backward_begin(ar)
        lnq = self.__dict__["_lnq"] = ops.LnReduceQueue()
        B = saved["B"]
        g = ar.grad_view
        x, mean, rstd, ypre = saved["final"]
        dy = dout.reshape(-1, self.num_features)
        dx = ops.layernorm_bwd(dy, x, mean, rstd, self.norm.weight.data, g(self.norm.weight), g(self.norm.bias),
                               y_pre=ypre, defer=lnq)
        ar.mark(self.norm.weight, self.norm.bias)
        # the bias-table gradients are accumulated with atomics: clear all of them in one launch
        # (24 separate fills were 24 x ~6 us on the critical path)
        ops.zero_batch([g(sv[0].attn.relative_position_bias_table) for sv in saved["blocks"]])
        bi = len(saved["blocks"])
        for li in range(len(self.layers) - 1, -1, -1):
            layer = self.layers[li]
            H, W = layer.input_resolution
            if layer.downsample is not None:
                ds = layer.downsample
                xin, xm, mean, rstd = saved["merges"][li]
                dxm = ops.gemm(dx, ar.compute(ds.reduction.weight), b_kmajor=True)
                ops.gemm(dx, xm, a_kmajor=True, b_kmajor=True, out=g(ds.reduction.weight), out_f32=True)
                dx = ops.layernorm_bwd(dxm, xin, mean, rstd, ds.norm.weight.data, g(ds.norm.weight), g(ds.norm.bias),
                                       merge=(H, W), defer=lnq)
                ar.mark(ds.reduction.weight, ds.norm.weight, ds.norm.bias)
            pre = None                               # dx * (DropPath scales of the consumer's MLP branch), when the producer wrote it
            for k in range(len(layer.blocks)):
                bi -= 1
                nxt = saved["blocks"][bi - 1] if k + 1 < len(layer.blocks) else None          # the producer's LayerNorm backward writes the DropPath-scaled branch gradient
                dx, pre = self._block_bwd(ar, saved["blocks"][bi], dx, B, pre, nxt[3] if (nxt is not None and len(nxt) == 6) else None)
        cols, x0, mean, rstd = saved["pe"]
        pe = self.patch_embed
        dx0 = ops.layernorm_bwd(dx, x0, mean, rstd, pe.norm.weight.data, g(pe.norm.weight), g(pe.norm.bias), defer=lnq)
        lnq.flush()
        if ops.NATIVE:
            ops.host().lnq_flush(ops.stream_int())
        ops.gemm(dx0, cols, a_kmajor=True, b_kmajor=True, out=g(pe.proj.weight), out_f32=True,
                 a_colsum=g(pe.proj.bias))
        ops.join_side(dx0.device)                # all weight gradients are complete before anyone reads them
        ar.mark(pe.norm.weight, pe.norm.bias, pe.proj.weight, pe.proj.bias)

    def _block_desc(self, ar, blk):
        """Raw device pointers of one block for the native host path, built once per arena."""
        key = ("swin_desc", id(blk))
        d = ar._views.get(key)
        if d is None:
            at, mlp = blk.attn, blk.mlp
            c, g = ar.compute, ar.grad_view
            dd = lambda p: p.data.data_ptr()
            w = [c(at.qkv.weight).data_ptr(), c(at.proj.weight).data_ptr(), c(mlp.fc1.weight).data_ptr(), c(mlp.fc2.weight).data_ptr()]
            # (ptr, bytes) of the next block's qkv and the previous block's fc2 weights (MvltGemm.prefetch, see bert.py)
            blocks = [b for _, b in self._blocks()]
            i = next(k for k, b in enumerate(blocks) if b is blk)
            nxt = c(blocks[i + 1].attn.qkv.weight) if i + 1 < len(blocks) else None
            prv = c(blocks[i - 1].mlp.fc2.weight) if i > 0 else None
            for t in (nxt, prv):
                w += [t.data_ptr(), t.numel() * t.element_size()] if t is not None else [0, 0]
            f = [dd(blk.norm1.weight), dd(blk.norm1.bias), dd(at.qkv.bias), dd(at.proj.bias), dd(at.relative_position_bias_table),
                 dd(blk.norm2.weight), dd(blk.norm2.bias), dd(mlp.fc1.bias), dd(mlp.fc2.bias)]
            gr = [g(p).data_ptr() for p in (blk.norm1.weight, blk.norm1.bias, at.qkv.weight, at.qkv.bias, at.proj.weight,
                                            at.proj.bias, at.relative_position_bias_table, blk.norm2.weight, blk.norm2.bias,
                                            mlp.fc1.weight, mlp.fc1.bias, mlp.fc2.weight, mlp.fc2.bias)]
            d = ar._views[key] = (w, f, gr)
        return d

    def _block_bwd_native(self, ar, sv, dx2, B, dy2_pre=None, s2_next=None):
        blk, saved, s1, s2, H, W = sv
        w2n, n2w = batched_window_maps(B, H, W, blk.window_size, blk.shift_size, dx2.device)
        w, f, gr = self._block_desc(ar, blk)
        at, mlp = blk.attn, blk.mlp
        out = ops.host().swin_block_bwd(dx2, saved, w, f, gr, [B, H, blk.dim, blk.num_heads, blk.shift_size, 0],
                                        [w2n.data_ptr(), n2w.data_ptr()], at.scale, 0 if s1 is None else s1.data_ptr(),
                                        0 if s2 is None else s2.data_ptr(), 0 if s2_next is None else s2_next.data_ptr(),
                                        dy2_pre if s2 is not None else None, ops.stream_int(), ops.side_int(dx2.device))
        dx0 = out[0]
        ar.mark(mlp.fc1.weight, mlp.fc1.bias, mlp.fc2.weight, mlp.fc2.bias, blk.norm1.weight, blk.norm1.bias,
                blk.norm2.weight, blk.norm2.bias, at.qkv.weight, at.qkv.bias, at.proj.weight, at.proj.bias,
                at.relative_position_bias_table)
        return dx0, (out[1] if len(out) > 1 else None)

    def _block_bwd(self, ar, sv, dx2, B, dy2_pre=None, s2_next=None):
        if len(sv) == 6:
            return self._block_bwd_native(ar, sv, dx2, B, dy2_pre, s2_next)
        (blk, x, mean1, rstd1, xn1w, qkv, ao, lse, x1, mean2, rstd2, xn2, h, a, s1, s2, H, W) = sv
        g = ar.grad_view
        C, nH, ws = blk.dim, blk.num_heads, blk.window_size
        Lt = H * W
        nW = (H // ws) * (W // ws)
        w2n, n2w = batched_window_maps(B, H, W, ws, blk.shift_size, dx2.device)
        at, mlp = blk.attn, blk.mlp
        # ---- critical path on the main stream: dgrads, LN backward, attention backward
        dy2 = ops.rows_transform(dx2, rowscale=(s2, Lt)) if s2 is not None else dx2
        dh = ops.gemm(dy2, ar.compute(mlp.fc2.weight), b_kmajor=True, mul_gelu_grad=h)
        dxn2 = ops.gemm(dh, ar.compute(mlp.fc1.weight), b_kmajor=True)
        dx1, dyw = ops.layernorm_bwd(dxn2, x1, mean2, rstd2, blk.norm2.weight.data, g(blk.norm2.weight),
                                     g(blk.norm2.bias), dres=dx2,
                                     branch=dict(rowmap=n2w, rowscale=(s1, Lt) if s1 is not None else None),
                                     defer=self.__dict__["_lnq"])
        dtab = g(at.relative_position_bias_table)          # zeroed for all blocks at once in _backward
        if C >= _SWIN_BWD_ONE and blk.shift_size in (0, 3) and ops.swin_wmsa2_bwd_parts(dyw.dtype, B, H, C, nH):
            # one launch, second design: dxn1w comes back as nH / 3 partial products that the LayerNorm backward below sums
            dqkv, dxn1w = ops.swin_wmsa2_bwd(dyw, qkv, lse, B, H, nH, blk.shift_size, ar.compute(at.proj.weight),
                                             ar.compute(at.qkv.weight), at.relative_position_bias_table.data, at.scale, dtab)
        elif _FUSED_WMSA_BWD and ops.swin_wmsa_bwd_supported(dyw.dtype, C, nH):
            # opt-in (MVLT_FUSED_WMSA_BWD=1, Python host path): proj dgrad + attention backward + qkv dgrad in one
            # launch.  Correct, but slower than the three launches at B=32 (profiles/r2_wmsa_pmc.md), hence not default.
            wpt = ar.compute(at.proj.weight).t().contiguous()
            wqt = ar.compute(at.qkv.weight).t().contiguous()
            dqkv, dxn1w = ops.swin_wmsa_bwd(dyw, qkv, lse, B, H, nH, blk.shift_size, wpt, wqt,
                                            at.relative_position_bias_table.data, at.scale, dtab)
        else:
            wp = ar.compute(at.proj.weight)
            if _SWIN_BWD_PROJ and ops.swin_bwd_proj_supported(dyw.dtype, nH, C // nH, blk.shift_size):
                dao, wp = dyw, wp               # the projection's dgrad rides inside the attention backward (MvltAttn.dout_weight)
            else:
                dao, wp = ops.gemm(dyw, wp, b_kmajor=True), None
            dqkv = ops.attn_bwd(dao, qkv, ao, lse, L.ATTN_SWIN, B * nW, ws * ws, nH, C // nH, at.scale,
                                dbias_table=dtab, bias_table=at.relative_position_bias_table.data, nW=nW, win_res=H,
                                shift=blk.shift_size, dout_weight=wp)
            dxn1w = ops.gemm(dqkv, ar.compute(at.qkv.weight), b_kmajor=True)
        dx0 = ops.layernorm_bwd(dxn1w, x, mean1, rstd1, blk.norm1.weight.data, g(blk.norm1.weight),
                                g(blk.norm1.bias), dy_rowmap=n2w, dres=dx1, defer=self.__dict__["_lnq"], dy_parts=dxn1w.dim() == 3)
        # ---- weight / bias gradients (only the optimizer consumes them): side stream, overlapping the next block
        with ops.on_side(dx2.device, dy2, a, dh, xn2, dyw, ao, dqkv, xn1w):
            ops.wgrad_group([(dy2, a, g(mlp.fc2.weight), g(mlp.fc2.bias)),
                             (dh, xn2, g(mlp.fc1.weight), g(mlp.fc1.bias)),
                             (dyw, ao, g(at.proj.weight), g(at.proj.bias)),
                             (dqkv, xn1w, g(at.qkv.weight), g(at.qkv.bias))])
        ar.mark(mlp.fc1.weight, mlp.fc1.bias, mlp.fc2.weight, mlp.fc2.bias, blk.norm1.weight, blk.norm1.bias,
                blk.norm2.weight, blk.norm2.bias, at.qkv.weight, at.qkv.bias, at.proj.weight, at.proj.bias,
                at.relative_position_bias_table)
        return dx0, None
