"""Flat parameter arena: memory laid out for one MI355X (288 GB HBM3E).

All parameters of a model live in ONE contiguous f32 buffer (``flat``); the
``nn.Parameter`` objects (reference state-dict names and shapes) are views
into it, so ``state_dict()`` / ``load_state_dict()`` / ``torch.save`` keep
working.  Alongside: one flat f32 gradient buffer (``grad``; ``p.grad`` are
views) and, in bf16 mode, one flat bf16 *compute copy* (``shadow``) that the
MFMA kernels read.  Consequences:

* query/key/value weights of a BERT layer are adjacent -> the fused QKV GEMM
  reads them as one [3H, H] matrix without a concat;
* AdamW is a handful of launches over contiguous ranges (one per run of
  parameters that received a gradient this step) and refreshes the bf16 copy
  in the same pass;
* gradient all-reduce buckets are plain slices of ``grad``.
"""
from __future__ import annotations

import operator
from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.nn as nn

ALIGN = 64  # elements (256 B for f32, 128 B for the bf16 copy)
_VERSION = operator.attrgetter("_version")


def _nothing():
    return None


def _mark_changed_hook(module, incompatible_keys):
    """load_state_dict post-hook, registered ONCE per module (a module-level function: picklable, and it holds
    no reference to any arena -- a rebuilt arena is found through the module, the old one can be freed)."""
    a = module.__dict__.get("_mvlt_arena")
    if a is not None:
        a.mark_parameters_changed()


class Arena:
    def __reduce__(self):
        # whole-module pickling (torch.save(model)): the arena is derived state keyed by object identity; it
        # unpickles as None and is rebuilt from the parameters on first use (Arena.of)
        return (_nothing, ())

    def __init__(self, root: nn.Module, compute_dtype: torch.dtype, allow_cpu: bool = False):
        # allow_cpu: host-logic tests only (bucketing / DDP over gloo); no kernel ever runs on CPU tensors
        params: List[Tuple[str, nn.Parameter]] = []
        seen = set()
        for name, p in root.named_parameters():
            if id(p) in seen:
                continue
            seen.add(id(p))
            params.append((name, p))
        if not params:
            raise RuntimeError("module has no parameters")
        dev = params[0][1].device
        if dev.type != "cuda" and not allow_cpu:
            raise RuntimeError("mvlt_amd modules run on the GPU only: call .cuda() first (no CPU fallback)")
        # fused groups (must be adjacent): declared by modules through _arena_groups()
        order: List[Tuple[str, nn.Parameter]] = []
        placed = set()
        group_of: Dict[int, List[nn.Parameter]] = {}
        for m in root.modules():
            for grp in getattr(m, "_arena_groups", lambda: [])():
                for p in grp:
                    group_of[id(p)] = grp
        names = {id(p): n for n, p in params}
        for name, p in params:
            if id(p) in placed:
                continue
            for q in group_of.get(id(p), [p]):
                if id(q) not in placed:
                    placed.add(id(q))
                    order.append((names[id(q)], q))
        self.names: List[str] = []
        self.offset: Dict[int, int] = {}
        self.numel: Dict[int, int] = {}
        self.params: List[nn.Parameter] = []
        off = 0
        for name, p in order:
            self.names.append(name)
            self.params.append(p)
            self.offset[id(p)] = off
            self.numel[id(p)] = p.numel()
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        self._starts: List[int] = [self.offset[id(p)] for p in self.params]      # ascending (bisect)
        self.device = dev
        self.compute_dtype = compute_dtype
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.shadow = torch.zeros(off, dtype=torch.bfloat16, device=dev) if compute_dtype == torch.bfloat16 else None
        with torch.no_grad():
            for p in self.params:
                o, n = self.offset[id(p)], p.numel()
                view = self.flat[o:o + n].view(p.shape)
                view.copy_(p.data.to(torch.float32))
                p.data = view
        self._ptr0 = self.params[0].data_ptr()
        self.shadow_fresh = False
        self._flat_version = -1
        # In-place writes through the Parameters (load_state_dict, a stock torch optimizer) bump THEIR version
        # counters, not ``flat``'s (``p.data = view`` does not share the counter), so staleness of the bf16 copy
        # is detected by the sum of the parameter versions -- scanned once per pass (after every backward, after
        # every load_state_dict; mark_parameters_changed() for anything else).
        self._param_versions = -1
        self._scan_ok = False
        for m in root.modules():
            if not m.__dict__.get("_mvlt_sd_hooked", False):
                m.register_load_state_dict_post_hook(_mark_changed_hook)
                m.__dict__["_mvlt_sd_hooked"] = True
        self._views: Dict[tuple, torch.Tensor] = {}
        self.has_grad = _EpochFlags(self.params)       # has_grad[id(p)] -> bool, reset in O(1) per backward pass
        self._marked: List[nn.Parameter] = []
        self._published: Optional[frozenset] = None
        self._ranges_cache: Dict[frozenset, list] = {}
        self.steps: Dict[int, int] = {id(p): 0 for p in self.params}
        self.exp_avg: Optional[torch.Tensor] = None
        self.exp_avg_sq: Optional[torch.Tensor] = None
        for m in root.modules():
            m.__dict__["_mvlt_arena"] = self

    # ------------------------------------------------------------------ lookup
    @staticmethod
    def of(module: nn.Module, compute_dtype: torch.dtype) -> "Arena":
        a = module.__dict__.get("_mvlt_arena")
        first = next(module.parameters())
        if (a is None or id(first) not in a.offset or a.compute_dtype != compute_dtype
                or first.data_ptr() != a.flat.data_ptr() + 4 * a.offset[id(first)]):
            a = Arena(module, compute_dtype)
        return a

    def master(self, p: nn.Parameter) -> torch.Tensor:
        return p.data

    def compute(self, p: nn.Parameter, rows: Optional[int] = None) -> torch.Tensor:
        """2-D compute-dtype view of a weight ([out, in]); ``rows`` > p.shape[0]
        spans the adjacent parameters of a fused group (QKV).  Views are cached:
        creating ~900 slice/view tensors per step is measurable host time."""
        key = ("c", id(p), rows)
        v = self._views.get(key)
        if v is None:
            v = self._views[key] = self._compute(p, rows)
        return v

    def _compute(self, p: nn.Parameter, rows: Optional[int] = None) -> torch.Tensor:
        o = self.offset[id(p)]
        cols = p.numel() // p.shape[0]
        r = p.shape[0] if rows is None else rows
        src = self.flat if self.shadow is None else self.shadow
        return src[o:o + r * cols].view(r, cols)

    def master_span(self, p: nn.Parameter, n: int) -> torch.Tensor:
        key = ("m", id(p), n)
        v = self._views.get(key)
        if v is None:
            o = self.offset[id(p)]
            v = self._views[key] = self.flat[o:o + n]
        return v

    def grad_view(self, p: nn.Parameter, rows: Optional[int] = None) -> torch.Tensor:
        key = ("g", id(p), rows)
        v = self._views.get(key)
        if v is None:
            v = self._views[key] = self._grad_view(p, rows)
        return v

    def _grad_view(self, p: nn.Parameter, rows: Optional[int] = None) -> torch.Tensor:
        o = self.offset[id(p)]
        if p.dim() == 1:
            n = p.numel() if rows is None else rows
            return self.grad[o:o + n]
        cols = p.numel() // p.shape[0]
        r = p.shape[0] if rows is None else rows
        return self.grad[o:o + r * cols].view(r, cols)

    def mark(self, *ps: nn.Parameter) -> None:
        """Gradients of ``ps`` have been written.  The backward pass walks the
        arena downwards, so everything at or above the lowest offset marked so
        far is final: that watermark drives the DDP bucket launches."""
        lo = None
        for p in ps:
            if self.has_grad[id(p)]:
                # gradients are written, not accumulated: a module that ran twice before one backward pass
                # (two forward calls whose losses are summed) would silently keep only the last contribution
                raise RuntimeError("mvlt_amd: a parameter received a second gradient in the same backward pass "
                                   "(the same module was run more than once before backward()); gradients are "
                                   "overwritten, not accumulated -- batch the inputs into one forward call instead")
            self.has_grad.set(id(p))
            self._marked.append(p)
            o = self.offset[id(p)]
            lo = o if lo is None or o < lo else lo
        cb = self.__dict__.get("_on_watermark")
        if cb is not None and lo is not None:
            wm = self.__dict__.get("_watermark", self.total)
            if lo < wm:
                self._watermark = lo
                cb(self, lo)

    # ------------------------------------------------------------------ per-step state
    def begin_backward(self) -> None:
        self._scan_ok = False          # an optimizer step usually follows: look at the parameter versions again
        self._stash_live_grads()
        self.has_grad.clear()
        self._marked = []
        self._watermark = self.total
        cb = self.__dict__.get("_on_backward_begin")
        if cb is not None:
            cb(self)

    def publish_grads(self) -> None:
        """Expose gradients the torch way: p.grad is a view for parameters that
        received a gradient this step and None for the others (so a stock
        torch optimizer skips them exactly as it does for the reference).
        A stock optimizer's ``zero_grad()`` (set_to_none=True, the reference loop run_pretrain.py:182-184) drops
        every view each step and the marked set changes with the seq2seq/bidir coin flip, so EVERY marked
        parameter whose ``grad`` is None gets its (cached) view back -- not only the ones whose status changed."""
        self._merge_stashed_grads()
        if self.grad.is_cuda:
            from . import ops
            if ops._side_keepalive:                  # weight gradients still queued on the side stream (a head's, when
                ops.join_side(self.grad.device)      # no backbone pass behind it has joined already)
        cur = frozenset(id(p) for p in self._marked)
        prev = self._published or frozenset()
        gv = self.__dict__.get("_pgrad_views")
        if gv is None:
            gv = self._pgrad_views = {}
        for p in self._marked:
            if p.grad is None:
                pid = id(p)
                v = gv.get(pid)
                if v is None:
                    o = self.offset[pid]
                    v = gv[pid] = self.grad[o:o + p.numel()].view(p.shape)
                p.grad = v
        if prev != cur:
            byid = self.__dict__.get("_byid")
            if byid is None:
                byid = self._byid = {id(p): p for p in self.params}
            for pid in prev - cur:
                byid[pid].grad = None
        self._published = cur
        self._consumed = False

    def note_grads_consumed(self) -> None:
        """FusedAdamW.step() / zero_grad(): the gradients of the last backward pass have been used."""
        self._consumed = True

    # ------------------------------------------------------------------ gradient accumulation (torch semantics)
    # The kernels WRITE parameter gradients.  torch accumulates into p.grad until it is cleared, so when a
    # backward pass starts while gradients of an earlier pass are still live (p.grad is not None and no
    # FusedAdamW.step()/zero_grad() in between -- a gradient-accumulation loop, or two losses backpropagated one
    # after the other) the live ranges are copied aside first and added back when the pass ends.  Two extra
    # streaming passes over the live ranges, paid only by loops that accumulate; the reference loop
    # (run_pretrain.py:181-184: backward, step, zero_grad) never does.
    def _stash_live_grads(self) -> None:
        self._acc_ranges, self._acc_params = [], []
        prev = self._marked
        if not prev or self.__dict__.get("_consumed", True):
            return
        live = [p for p in prev if p.grad is not None]
        if not live:
            return
        live.sort(key=lambda p: self.offset[id(p)])
        ranges: List[List[int]] = []
        for p in live:
            o = self.offset[id(p)]
            e = o + (p.numel() + ALIGN - 1) // ALIGN * ALIGN
            if ranges and ranges[-1][1] == o:
                ranges[-1][1] = e
            else:
                ranges.append([o, e])
        st = self.__dict__.get("_stash")
        if st is None:
            st = self._stash = torch.empty_like(self.grad)
        for a, b in ranges:
            st[a:b].copy_(self.grad[a:b])
        self._acc_ranges, self._acc_params = ranges, live

    def _merge_stashed_grads(self) -> None:
        if not self.__dict__.get("_acc_ranges"):
            return
        if self.grad.is_cuda:                        # weight gradients of this pass come from the side stream
            from . import ops
            ops.join_side(self.grad.device)
        for p in self._acc_params:                   # a parameter only the earlier pass touched keeps that gradient
            pid = id(p)
            if not self.has_grad[pid]:
                o = self.offset[pid]
                self.grad[o:o + p.numel()].zero_()
                self.has_grad.set(pid)
                self._marked.append(p)
        for a, b in self._acc_ranges:
            self.grad[a:b].add_(self._stash[a:b])
        self._acc_ranges, self._acc_params = [], []

    def mark_parameters_changed(self) -> None:
        """Call after writing parameter values by any route the arena cannot see."""
        self._scan_ok = False

    def refresh_shadow(self, tail: bool = True) -> None:
        """bf16 compute copy <- f32 master (one pass, 6 B/param) whenever the master was modified through
        torch: in-place ops on ``flat`` itself bump ``flat._version``; writes through the Parameters
        (load_state_dict, a stock torch optimizer) are seen through the parameter version scan.  The fused
        AdamW kernel refreshes the copy itself and leaves all versions untouched.
        ``tail``: a forward pass is about to read parameters; an optimizer tail that FusedAdamW deferred (optim.py) is
        applied first, in stream order.  Only the two callers that handle the tail themselves pass False: the Swin forward
        (its parameters are never deferred) and MVLBert._forward (which launches the tail beside the encoder)."""
        if tail and self.__dict__.get("_opt_tail") is not None:
            self._opt_tail.flush()
        self._in_backward = False        # a forward pass: any earlier backward pass is over (also an aborted one)
        if self.shadow is None:
            return
        if not self._scan_ok:
            pv = sum(map(_VERSION, self.params))
            if pv != self._param_versions:
                self._param_versions = pv
                self.shadow_fresh = False
            self._scan_ok = True
        if not self.shadow_fresh or self.flat._version != self._flat_version:
            from . import ops
            ops.cast(self.flat, torch.bfloat16, out=self.shadow)
            self.shadow_fresh = True
            self._flat_version = self.flat._version

    def note_params_written_by_kernel(self) -> None:
        self._flat_version = self.flat._version

    def active_ranges(self) -> List[Tuple[int, int, int]]:
        """Maximal contiguous [start, end) element ranges of parameters that
        have a gradient and share the same optimizer step count."""
        out: List[List[int]] = []
        cur = None
        for p in self.params:
            pid = id(p)
            if not self.has_grad[pid]:
                cur = None
                continue
            o = self.offset[pid]
            e = o + (p.numel() + ALIGN - 1) // ALIGN * ALIGN
            st = self.steps[pid]
            if cur is not None and cur[1] == o and cur[2] == st:
                cur[1] = e
            else:
                cur = [o, e, st]
                out.append(cur)
        return [(a, b, c) for a, b, c in out]

    def params_between(self, lo: int, hi: int):
        """Parameters that lie entirely inside the element range [lo, hi), in arena order."""
        import bisect
        i = bisect.bisect_left(self._starts, lo)
        while i < len(self.params) and self._starts[i] < hi:
            p = self.params[i]
            if self._starts[i] + (p.numel() + ALIGN - 1) // ALIGN * ALIGN <= hi:
                yield p
            i += 1

    def bump_steps(self) -> None:
        for p in self._marked:
            self.steps[id(p)] += 1


class _EpochFlags:
    """dict-like bool flags keyed by id(param) with an O(1) clear()."""

    def __init__(self, params):
        self.epoch = 1
        self.stamp = {id(p): 0 for p in params}

    def __getitem__(self, pid):
        return self.stamp[pid] == self.epoch

    def set(self, pid):
        self.stamp[pid] = self.epoch

    def clear(self):
        self.epoch += 1
