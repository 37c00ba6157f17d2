"""Pre-training step driver (reference loop: run_pretrain.py:162-194) and the
synthetic batch generator of SURVEY.md section 8(d)."""
import random

import torch

from . import ops
from .optim import FusedAdamW


def synthetic_batch(B, T, device, seed, vocab=30522, itm=True, with_lengths=False, lengths=None):
    """image N(0,1) [B,3,224,224]; caption ids U{1000..vocab-1}, last real id = [END]=104,
    zero padded; <=10 MLM labels per sample (20%), 80% of them replaced by [MASK]=103;
    ITM labels Bernoulli(0.5) (run_pretrain_rgc_roco_medicat.py:134-212).  ``lengths``: caption lengths to use instead of
    drawing them (a data-parallel run deals the global batch's lengths to its ranks: data.deal_balanced)."""
    g = torch.Generator().manual_seed(seed)
    image = torch.randn(B, 3, 224, 224, generator=g)
    ids = torch.zeros(B, T, dtype=torch.long)
    labels = torch.full((B, T), -100, dtype=torch.long)
    for b in range(B):
        ln = int(torch.randint(16, T, (1,), generator=g))
        if lengths is not None:
            ln = int(lengths[b])
        row = torch.randint(1000, vocab, (ln,), generator=g)
        row[-1] = 104
        nm = min(10, max(1, round(0.2 * ln)))
        pos = torch.randperm(ln, generator=g)[:nm]
        labels[b, pos] = row[pos]
        row[pos[: max(1, int(0.8 * nm))]] = 103
        ids[b, :ln] = row
    itm_l = torch.randint(0, 2, (B,), generator=g) if itm else torch.ones(B, dtype=torch.long)
    out = tuple(t.to(device) for t in (image, ids, labels, itm_l))
    if with_lengths:        # caption lengths as the tokeniser knows them (host side): enables packed rows
        out += ((ids != 0).sum(1).to(torch.int32),)
    return out


class PretrainStep:
    """loss = model(batch); loss.backward(); optimizer.step()  -- one call per step.
    ``reducer`` (mvlt_amd.ddp.GradReducer) makes it data parallel."""

    def __init__(self, model, lr=None, reducer=None, world_size=1, overlap_optimizer=False, defer_optimizer_tail=False):
        """defer_optimizer_tail (opt-in): the AdamW sweep over BertLayers 1.., pooler and heads is left to the NEXT call, which
        runs it beside its encoder forward (optim.FusedAdamW(defer_tail=True)); between two calls those parameters are one
        update behind until something reads them -- a forward pass, ``flush()``, state_dict() / save_pretrained() all apply it."""
        self.model = model
        # the reducer hands out rank-averaged gradients by default (like torch DDP); a SUM reducer is rescaled here
        gs = 1.0 if (reducer is None or getattr(reducer, "average", False)) else 1.0 / world_size
        self.opt = FusedAdamW(model, lr=lr if lr is not None else model.config.lr, betas=(0.9, 0.999), eps=1e-6,
                              weight_decay=1e-4, grad_scale=gs, defer_tail=defer_optimizer_tail and not overlap_optimizer)
        self.reducer = reducer
        if overlap_optimizer:
            # opt-in: AdamW of finished arena slices is queued beside the rest of the backward pass.  On one
            # GPU the step is throughput-bound and this measured neutral (18.8 vs 19.0 ms); it exists for DDP,
            # where it takes all but the last bucket's update off the serial tail behind the all-reduce.
            self.opt.overlap_with_backward(reducer)

    def __call__(self, batch):
        # batch = (image, caption_masked, caption_label, image_text_label[, text_lengths (host ints)])
        loss = self.model(*batch[:4], text_lengths=batch[4]) if len(batch) > 4 else self.model(*batch)
        loss.backward()
        self.opt.step()
        # in-launch hand-offs (mvlt_swin_wmsa2_fwd) that timed out poison the loss with NaN AND raise here, one step late:
        # the error counts travel to pinned host memory behind the step's kernels, no device sync (ops.wmsa2_check)
        ops.wmsa2_check(sync=False)
        if self.reducer is not None:
            # data parallel: the error count of ALL ranks (it travelled with the label count): every rank raises, not only the one
            # whose wait ran out.  By now the NaN gradients of that step have been applied -- reload the last checkpoint.
            self.reducer.check_handoff()
        return loss

    def flush(self):
        """Apply a deferred optimizer tail (end of training, before parameters are read through torch)."""
        self.opt.flush()
