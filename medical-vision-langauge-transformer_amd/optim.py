"""Fused AdamW over the flat parameter arena (reference: torch.optim.AdamW at
run_pretrain.py:165-166 -- lr 4e-5, betas (0.9, 0.999), eps 1e-6, weight_decay 1e-4).

One ``mvlt_adamw`` launch per contiguous run of parameters that received a
gradient this step (typically 4-6 launches for 208.9 M parameters); the same
pass refreshes the bf16 compute copy.  Parameters without a gradient are
skipped entirely -- no weight decay, no moment update, own step counter --
exactly like torch.optim.AdamW treats ``p.grad is None`` (idle MLM head,
unused ``head`` / ``resnet_fc`` / ``embedding_LayerNorm``).

``overlap_with_backward``: AdamW is HBM-bound (30 B/parameter) while the
backward pass is MFMA-bound, so the update of a finished slice of the arena
(everything above the backward watermark; under DDP: a bucket whose
all-reduce has been issued) is queued on a third HIP stream while the
backward pass continues below it.  ``step()`` then only handles what is left
and joins the stream.  Same arithmetic, same result; only for loops that call
``backward(); step()`` once each per batch (mvlt_amd.train.PretrainStep).
"""
import torch

from . import ops
from .arena import ALIGN, Arena
from .runtime import compute_dtype_of


class _OptTail:
    """The part of one optimizer step that FusedAdamW(defer_tail=True) has NOT launched yet: the runs of the arena above
    ``split`` (BertLayers 1.. and the heads), in chunks.  The next forward pass launches them on the optimizer stream
    beside the encoder (``launch``: MVLBert._forward, behind the Swin tower) and waits for a chunk right before the first
    layer that reads its parameters (``wait_for``); anything else that is about to read parameters applies what is left in
    stream order (``flush``: Arena.refresh_shadow, the next step(), state_dict, PretrainStep.flush)."""

    def __init__(self, opt, ar, chunks):
        self.ar = ar
        self.chunks = chunks            # [dict(lo=, runs=[(lo, hi, step)], event=None, waited=False)] in arena order
        self.launched = False
        # the hyper-parameters of the step this tail belongs to (an LR schedule may move opt.lr before the tail runs)
        self.hyper = (opt.lr, opt.betas[0], opt.betas[1], opt.eps, opt.weight_decay, opt.grad_scale)

    def _run(self, ch):
        lr, b1, b2, eps, wd, gs = self.hyper
        ar = self.ar
        for lo, hi, st in ch["runs"]:
            ops.adamw(ar.flat[lo:hi], ar.grad[lo:hi], ar.exp_avg[lo:hi], ar.exp_avg_sq[lo:hi],
                      ar.shadow[lo:hi] if ar.shadow is not None else None, lr, b1, b2, eps, wd, st + 1, gs)

    def launch(self):
        """Queue every chunk on the optimizer stream behind what the current stream has queued so far."""
        if self.launched:
            return
        self.launched = True
        dev = self.ar.flat.device
        st = ops.opt_stream(dev)
        st.wait_stream(torch.cuda.current_stream())
        with ops.on_stream(st, "opt"):
            for ch in self.chunks:
                self._run(ch)
                ch["event"] = torch.cuda.Event()
                ch["event"].record(st)

    def wait_for(self, offset=None):
        """The current stream waits for every chunk that holds parameters below ``offset`` (None: all of them)."""
        done = True
        for ch in self.chunks:
            if ch["waited"]:
                continue
            if offset is None or ch["lo"] < offset:
                torch.cuda.current_stream().wait_event(ch["event"])
                ch["waited"] = True
            else:
                done = False
        if done:
            self.ar.__dict__["_opt_tail"] = None

    def flush(self):
        """Apply what has not been launched in stream order on the CURRENT stream; wait for what has."""
        if not self.launched:
            self.launched = True
            for ch in self.chunks:
                self._run(ch)
                ch["waited"] = True
            self.ar.__dict__["_opt_tail"] = None
        else:
            self.wait_for(None)


class FusedAdamW:
    def __init__(self, model, lr=4e-5, betas=(0.9, 0.999), eps=1e-6, weight_decay=1e-4, grad_scale=1.0, defer_tail=False):
        self.model = model
        # defer_tail (opt-in, mvlt_amd.train.PretrainStep(defer_optimizer_tail=True)): step() updates the parameters the next
        # forward pass reads FIRST (Swin tower, embeddings, BertLayer 0) and leaves the rest -- BertLayers 1.., pooler, heads:
        # 102 M of the 182 M updated parameters, 3.1 GB of the sweep's 5.5 GB -- to the next forward pass, which runs them on the
        # optimizer stream BESIDE the encoder: the sweep is HBM-bound (6 TB/s, no LDS, 57 registers), the BertLayer forward is
        # bound by its GEMMs' L2 -> LDS traffic and moves < 1 TB/s of HBM bytes.  (Not beside the Swin tower: its fused W-MSA
        # workgroups take a whole CU's registers and would queue behind the sweep's waves.)  Same arithmetic, same result:
        # tests/test_model_gpu.py::test_deferred_optimizer_tail_equals_plain_steps.
        self.defer_tail = bool(defer_tail)
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.grad_scale = grad_scale           # 1/world_size under DDP (gradients are summed, not averaged)
        self._arena = None
        self._overlap = None                   # dict(reducer=, chunk=) when enabled
        self._stepped = set()                  # id(param) already updated during the current backward pass
        self._pending_hi = 0

    def _state(self) -> Arena:
        ar = Arena.of(self.model, compute_dtype_of(self.model))
        if ar is not self._arena or ar.exp_avg is None:
            ar.exp_avg = torch.zeros_like(ar.flat)
            ar.exp_avg_sq = torch.zeros_like(ar.flat)
            old = self._arena
            if old is not None and old is not ar and old.exp_avg is not None:
                # the arena was rebuilt (dtype switch, .cuda()/.to() after first use): carry the moments and step
                # counts over, parameter by parameter, instead of silently restarting Adam
                for p in ar.params:
                    pid = id(p)
                    if pid in old.offset and old.numel[pid] == p.numel():
                        o, oo, n = ar.offset[pid], old.offset[pid], p.numel()
                        ar.exp_avg[o:o + n].copy_(old.exp_avg[oo:oo + n])
                        ar.exp_avg_sq[o:o + n].copy_(old.exp_avg_sq[oo:oo + n])
                        ar.steps[pid] = old.steps[pid]
                old.exp_avg = old.exp_avg_sq = None
            self._arena = ar
            self.__dict__["_plans"] = {}          # cached (lo, hi) sweep ranges are offsets into the OLD arena layout
            if self._overlap is not None:
                self._hook(ar)
        return ar

    def zero_grad(self, set_to_none=True):
        """Gradients are overwritten (not accumulated) by every backward pass and
        parameters without a gradient are tracked per step, so nothing to clear; the call only tells the
        arena that the caller is done with the last backward pass's gradients."""
        if self._arena is not None:
            self._arena.note_grads_consumed()

    # ------------------------------------------------------------------ checkpoint / resume (absent in the reference)
    def state_dict(self):
        """Moments and per-parameter step counts keyed by parameter NAME (independent of the arena layout)."""
        ar = self._state()
        self.flush()
        out = {"hyper": dict(lr=self.lr, betas=tuple(self.betas), eps=self.eps, weight_decay=self.weight_decay),
               "state": {}}
        for name, p in zip(ar.names, ar.params):
            o, n = ar.offset[id(p)], p.numel()
            if ar.steps[id(p)] > 0:
                out["state"][name] = dict(step=ar.steps[id(p)],
                                          exp_avg=ar.exp_avg[o:o + n].view(p.shape).detach().cpu().clone(),
                                          exp_avg_sq=ar.exp_avg_sq[o:o + n].view(p.shape).detach().cpu().clone())
        return out

    def load_state_dict(self, sd):
        ar = self._state()
        self.flush()
        hp = sd.get("hyper", {})
        self.lr = hp.get("lr", self.lr)
        self.betas = tuple(hp.get("betas", self.betas))
        self.eps = hp.get("eps", self.eps)
        self.weight_decay = hp.get("weight_decay", self.weight_decay)
        byname = dict(zip(ar.names, ar.params))
        unknown = [k for k in sd["state"] if k not in byname]
        if unknown:
            raise KeyError(f"optimizer state for unknown parameters: {unknown[:5]}")
        ar.exp_avg.zero_(); ar.exp_avg_sq.zero_()
        for p in ar.params:
            ar.steps[id(p)] = 0
        for name, st in sd["state"].items():
            p = byname[name]
            o, n = ar.offset[id(p)], p.numel()
            ar.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
            ar.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            ar.steps[id(p)] = int(st["step"])

    # ------------------------------------------------------------------ update of a set of element ranges
    def _apply(self, ar: Arena, params, extra_scale: float = 1.0) -> None:
        """AdamW over ``params`` (arena order): one launch per contiguous run with equal step count."""
        b1, b2 = self.betas
        run = None
        runs = []
        for p in params:
            pid = id(p)
            o = ar.offset[pid]
            e = o + (p.numel() + ALIGN - 1) // ALIGN * ALIGN
            st = ar.steps[pid]
            if run is not None and run[1] == o and run[2] == st:
                run[1] = e
            else:
                run = [o, e, st]
                runs.append(run)
        for lo, hi, st in runs:
            ops.adamw(ar.flat[lo:hi], ar.grad[lo:hi], ar.exp_avg[lo:hi], ar.exp_avg_sq[lo:hi],
                      ar.shadow[lo:hi] if ar.shadow is not None else None,
                      self.lr, b1, b2, self.eps, self.weight_decay, st + 1, self.grad_scale * extra_scale)

    def _plan_runs(self, ar: Arena, params):
        """[(lo, hi, ids of the parameters of the run)]: maximal contiguous arena ranges of ``params`` that
        currently share one step count."""
        runs, run = [], None
        for p in params:
            pid = id(p)
            o = ar.offset[pid]
            e = o + (p.numel() + ALIGN - 1) // ALIGN * ALIGN
            st = ar.steps[pid]
            if run is not None and run[1] == o and run[3] == st:
                run[1] = e
                run[2].append(pid)
            else:
                run = [o, e, [pid], st]
                runs.append(run)
        return [(lo, hi, tuple(ids)) for lo, hi, ids, _ in runs]

    # ------------------------------------------------------------------ overlap with the backward pass
    def overlap_with_backward(self, reducer=None, chunk_bytes: int = 64 << 20) -> "FusedAdamW":
        self._overlap = dict(reducer=reducer, chunk=chunk_bytes // 4)
        self._hook(self._state())
        return self

    def _hook(self, ar: Arena) -> None:
        red = self._overlap["reducer"]
        if red is not None:
            red.on_bucket = self._on_bucket
            prev_begin = red._begin

            def begin(arena, prev_begin=prev_begin):
                prev_begin(arena)
                self._begin(arena)
            ar._on_backward_begin = begin
        else:
            ar._on_backward_begin = self._begin
            ar._on_watermark = self._on_watermark

    def _begin(self, ar: Arena) -> None:
        self._stepped = set()
        self._pending_hi = ar.total

    def _on_watermark(self, ar: Arena, lo: int) -> None:
        # single GPU: everything at or above the watermark has its final gradient queued
        if self._pending_hi - lo < self._overlap["chunk"]:
            return
        hi, self._pending_hi = self._pending_hi, lo
        params = [p for p in ar.params_between(lo, hi) if ar.has_grad[id(p)] and id(p) not in self._stepped]
        if not params:
            return
        ops.LnReduceQueue.flush_all()                       # deferred LayerNorm gamma/beta gradients
        st = ops.opt_stream(ar.device)
        st.wait_stream(torch.cuda.current_stream())         # dgrad chain + flushed reductions
        st.wait_stream(ops.side_stream(ar.device))          # weight gradients
        with ops.on_stream(st, "opt"):
            self._apply(ar, params)
        self._stepped.update(id(p) for p in params)

    def _on_bucket(self, ar: Arena, ranges, handles, owed_scale: float = 1.0) -> None:
        # DDP: the bucket's all-reduce has been issued; the update waits for it on the optimizer stream
        params = [p for a, b in ranges for p in ar.params_between(a, b)
                  if ar.has_grad[id(p)] and id(p) not in self._stepped]
        if not params or not ar.flat.is_cuda:
            return
        st = ops.opt_stream(ar.device)
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            for h in handles:
                h.wait()
        with ops.on_stream(st, "opt"):
            self._apply(ar, params, owed_scale)
        self._stepped.update(id(p) for p in params)

    # ------------------------------------------------------------------ the torch.optim-style entry point
    def flush(self):
        """Apply a deferred optimizer tail now (in stream order).  No-op otherwise."""
        ar = self._arena
        if ar is not None and ar.__dict__.get("_opt_tail") is not None:
            ar._opt_tail.flush()

    def _tail_bounds(self, ar: Arena):
        """Arena offsets that cut the deferred tail into chunks: [BertLayer 1, BertLayer 4, BertLayer 8, behind the last
        BertLayer]; None when the model has no such layers (nothing is deferred then)."""
        key = id(ar)
        cached = self.__dict__.get("_tail_cache")
        if cached is not None and cached[0] == key:
            return cached[1]
        first = {}
        last_enc = 0
        for name, p in zip(ar.names, ar.params):
            k = name.find("encoder.layer.")
            if k >= 0:
                i = int(name[k + 14:].split(".")[0])
                o = ar.offset[id(p)]
                first[i] = min(first.get(i, o), o)
                last_enc = max(last_enc, o + (p.numel() + ALIGN - 1) // ALIGN * ALIGN)
        bounds = None
        if len(first) >= 2 and all(first[i] < first[i + 1] for i in range(len(first) - 1)):
            n = len(first)
            cuts = sorted({1, min(4, n - 1), min(8, n - 1)} - {0})
            bounds = [first[c] for c in cuts] + [last_enc]
        self._tail_cache = (key, bounds)
        return bounds

    @torch.no_grad()
    def step(self):
        ar = self._state()
        self.flush()                     # (a tail no forward pass has picked up: apply it before this step's update)
        if self._overlap is None:
            # the set of parameters with a gradient takes two values in pre-training (seq2seq / bidirectional head):
            # the contiguous runs are planned once per set, not rebuilt from 582 parameters every step
            key = ar._published
            plan = self.__dict__.setdefault("_plans", {}).get(key) if key is not None else None
            if plan is None:
                params = [p for p in ar.params if ar.has_grad[id(p)]]
                plan = self._plan_runs(ar, params)
                if key is not None and len(self._plans) < 16:
                    self._plans[key] = plan
            steps = ar.steps
            if not all(steps[ids[0]] == steps[i] for _, _, ids in plan for i in ids):
                # step counts inside a planned run have drifted apart (a parameter that belongs to several sets, a
                # resumed checkpoint): plan again for the current counts
                plan = self._plan_runs(ar, [p for p in ar.params if ar.has_grad[id(p)]])
                if key is not None:
                    self._plans[key] = plan
            b1, b2 = self.betas
            bounds = self._tail_bounds(ar) if (self.defer_tail and ar.flat.is_cuda) else None
            tail = []                    # (lo, hi, step) pieces above the split, in arena order
            for lo, hi, ids in plan:
                st = steps[ids[0]]
                if bounds is not None and hi > bounds[0]:
                    cut = max(lo, bounds[0])
                    tail.append((cut, hi, st))
                    hi = cut
                if hi > lo:
                    ops.adamw(ar.flat[lo:hi], ar.grad[lo:hi], ar.exp_avg[lo:hi], ar.exp_avg_sq[lo:hi],
                              ar.shadow[lo:hi] if ar.shadow is not None else None,
                              self.lr, b1, b2, self.eps, self.weight_decay, st + 1, self.grad_scale)
            if tail:
                edges = bounds + [ar.total]
                chunks = []
                for c in range(len(edges) - 1):
                    a, b = edges[c], edges[c + 1]
                    runs = [(max(lo, a), min(hi, b), st) for lo, hi, st in tail if min(hi, b) > max(lo, a)]
                    if runs:
                        chunks.append(dict(lo=a, runs=runs, event=None, waited=False))
                ar.__dict__["_opt_tail"] = _OptTail(self, ar, chunks)
        else:
            self._apply(ar, [p for p in ar.params if ar.has_grad[id(p)] and id(p) not in self._stepped])
            torch.cuda.current_stream().wait_stream(ops.opt_stream(ar.device))
            self._stepped = set()
        ar.bump_steps()
        ar.note_params_written_by_kernel()
        ar.note_grads_consumed()
