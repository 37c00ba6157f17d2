"""Integer tables of the hot path (bit-exact rows a2-a5, a13 of SURVEY.md section 8).

Only small host-side tables live here (built once, cached on the device); the
shift mask and relative-position index are *also* recomputed inside the
attention kernel (csrc/attn.hip) -- these copies exist as state-dict buffers
and for the parity tests against the reference's tables.
"""
from functools import lru_cache

import torch


def relative_position_index(ws: int) -> torch.Tensor:
    """[ws*ws, ws*ws] int64, (dy+ws-1)*(2ws-1) + (dx+ws-1)
    (reference: visual_feature_extractor.py:203-213)."""
    t = torch.arange(ws * ws)
    y, x = t // ws, t % ws
    return (y[:, None] - y[None, :] + ws - 1) * (2 * ws - 1) + (x[:, None] - x[None, :] + ws - 1)


def shift_attn_mask(H: int, W: int, ws: int, shift: int) -> torch.Tensor:
    """[nW, ws*ws, ws*ws] f32 0/-100 (reference: visual_feature_extractor.py:318-344)."""
    def band(n):
        r = torch.arange(n)
        return (r >= n - ws).long() + (r >= n - shift).long()
    reg = band(H)[:, None] * 3 + band(W)[None, :]
    reg = reg.view(H // ws, ws, W // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    return (reg[:, None, :] != reg[:, :, None]).float() * -100.0


def window_token_map(H: int, W: int, ws: int, shift: int) -> torch.Tensor:
    """win2nat[w*ws*ws + s]: token (h*W+w) of the un-shifted image sitting at
    window w, slot s after roll(-shift) + window_partition
    (reference: visual_feature_extractor.py:144-156, :360-367)."""
    h = (torch.arange(H) + shift) % H
    w = (torch.arange(W) + shift) % W
    src = h[:, None] * W + w[None, :]
    return src.view(H // ws, ws, W // ws, ws).permute(0, 2, 1, 3).reshape(-1)


def patch_merge_map(H: int, W: int) -> torch.Tensor:
    """[H/2*W/2, 4] source tokens in concat order x0,x1,x2,x3
    (reference: visual_feature_extractor.py:435-439)."""
    i = torch.arange(H // 2)[:, None]
    j = torch.arange(W // 2)[None, :]
    base = (2 * i) * W + 2 * j
    return torch.stack([base, base + W, base + 1, base + W + 1], -1).view(-1, 4)


_cache = {}


def batched_window_maps(B: int, H: int, W: int, ws: int, shift: int, device):
    """(win2nat, nat2win) int32 [B*H*W] over batch-flattened rows."""
    key = (B, H, W, ws, shift, str(device))
    hit = _cache.get(key)
    if hit is None:
        src = window_token_map(H, W, ws, shift)
        L = H * W
        w2n = (torch.arange(B)[:, None] * L + src[None, :]).reshape(-1)
        n2w = torch.empty_like(w2n)
        n2w[w2n] = torch.arange(B * L)
        hit = (w2n.to(torch.int32).to(device), n2w.to(torch.int32).to(device))
        _cache[key] = hit
    return hit
