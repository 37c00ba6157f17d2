"""Fused AdamW over the flat parameter arena (reference: torch.optim.AdamW at
run_pretrain.py:165-166 -- lr 4e-5, betas (0.9, 0.999), eps 1e-6, weight_decay 1e-4).

One ``mvlt_adamw`` launch per contiguous run of parameters that received a
gradient this step (typically 4-6 launches for 208.9 M parameters); the same
pass refreshes the bf16 compute copy.  Parameters without a gradient are
skipped entirely -- no weight decay, no moment update, own step counter --
exactly like torch.optim.AdamW treats ``p.grad is None`` (idle MLM head,
unused ``head`` / ``resnet_fc`` / ``embedding_LayerNorm``).
"""
import torch

from . import ops
from .arena import Arena
from .runtime import compute_dtype_of


class FusedAdamW:
    def __init__(self, model, lr=4e-5, betas=(0.9, 0.999), eps=1e-6, weight_decay=1e-4, grad_scale=1.0):
        self.model = model
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.grad_scale = grad_scale           # 1/world_size under DDP (gradients are summed, not averaged)
        self._arena = None

    def _state(self) -> Arena:
        ar = Arena.of(self.model, compute_dtype_of(self.model))
        if ar is not self._arena or ar.exp_avg is None:
            ar.exp_avg = torch.zeros_like(ar.flat)
            ar.exp_avg_sq = torch.zeros_like(ar.flat)
            self._arena = ar
        return ar

    def zero_grad(self, set_to_none=True):
        """Gradients are overwritten (not accumulated) by every backward pass and
        parameters without a gradient are tracked per step, so nothing to clear."""

    @torch.no_grad()
    def step(self):
        ar = self._state()
        b1, b2 = self.betas
        for lo, hi, st in ar.active_ranges():
            ops.adamw(ar.flat[lo:hi], ar.grad[lo:hi], ar.exp_avg[lo:hi], ar.exp_avg_sq[lo:hi],
                      ar.shadow[lo:hi] if ar.shadow is not None else None,
                      self.lr, b1, b2, self.eps, self.weight_decay, st + 1, self.grad_scale)
        ar.bump_steps()
        ar.note_params_written_by_kernel()
