"""Data-parallel pre-training over RCCL / xGMI: one process per GPU.

The reference has no distributed code; this is the new
exchange step BASELINE.json asks for.  Design for 8 x MI355X on one node
(xGMI is point-to-point, 7 links x ~153 GB/s per GPU):

* every rank holds a full replica (0.84 GB f32 + optimizer state: trivial in
  288 GB), the global batch is sharded across ranks, no data-path collective;
* the only collective is a SUM all-reduce of the flat f32 gradient buffer,
  issued in large contiguous buckets (default 64 MiB -- few, large transfers
  keep RCCL's rings/trees bandwidth-bound rather than latency-bound) as soon
  as the backward pass has finished the parameters of a bucket.  The backward
  walks the arena from its end to its start (heads -> BERT 11..0 ->
  embeddings -> Swin 3..0 -> patch embed), so readiness is a descending
  watermark and buckets are plain slices of ``arena.grad``;
* ``dist.all_reduce(async_op=True)`` orders itself after the kernels already
  queued on the compute stream and runs on RCCL's own stream, i.e. it
  overlaps the rest of the backward; the end-of-backward hook waits on the
  outstanding handles; AdamW divides by world_size (grad_scale);
* parameters without a gradient this step (idle MLM head, unused modules) are
  not communicated;
* the MLM loss of the reference is a mean over the labelled tokens of the WHOLE
  batch (model.py:410).  Ranks hold different numbers of labelled tokens, so the
  rank-average of per-rank means is a different number; ``global_label_mean``
  (default on) all-reduces the 4-byte label count in the forward pass and every
  rank divides its summed loss by N_global / world instead of its own count
  (only in forwards that record a graph: torch.no_grad() forwards and
  ``GradReducer.no_label_sync()`` take no collective and keep the per-rank mean;
  heads other than MVLBertForPretraining always use per-rank means):
  the averaged loss and gradients then equal the single-process global-batch
  step exactly (the ITM loss is a mean over samples, equal per rank: unchanged);
* the seq2seq/bidirectional coin flip of MVLBertForPretraining is drawn from
  Python's ``random``: ``seed_coin_flip`` gives all ranks the same stream so
  the same MLM head is active everywhere (otherwise buckets would diverge).
"""
from __future__ import annotations

import os
import random
from typing import List, Tuple

import torch
import torch.distributed as dist

from .arena import ALIGN, Arena
from .runtime import compute_dtype_of


def seed_coin_flip(seed: int) -> None:
    random.seed(seed)


GAP_ELEMS = 2 << 20      # ranges of one bucket separated by less than this (8 MB of f32) leave as ONE collective


def plan_ranges(arena: Arena, lo: int, hi: int, done: set, gap_elems: int = GAP_ELEMS, idle_ok=None,
                rode_along: list | None = None) -> List[Tuple[int, int]]:
    """Contiguous element ranges inside [lo, hi) covered by parameters that
    received a gradient this step and have not been reduced yet (``done`` is
    updated).  Ranges separated only by a SMALL run of parameters without a gradient (the never-used Swin classifier
    head, resnet_fc, embedding_LayerNorm: 9 MB in all) are merged, so a bucket is one collective; the gap's stale
    gradient slots are reduced along (nobody reads them: p.grad is None there).  The idle MLM head (96 MB) stays a gap.
    A gap is merged only when ``idle_ok(p)`` holds for every parameter in it (GradReducer: frozen, or without a gradient
    in the previous completed step -- a parameter whose gradient may still arrive later in this backward pass must not
    be swept into an in-place collective); the ids of the parameters that ride along are appended to ``rode_along``."""
    out: List[List[int]] = []
    gap: List[int] = []                      # ids of the gradient-less parameters since the last range
    gap_ok = True
    for p in arena.params_between(lo, hi):
        o = arena.offset[id(p)]
        e = o + (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        if not arena.has_grad[id(p)] or id(p) in done:
            if not arena.has_grad[id(p)]:
                gap.append(id(p))
                gap_ok = gap_ok and (idle_ok is None or idle_ok(p))
            else:
                gap_ok = False               # reduced earlier in this pass: never ride along a second time
            continue
        done.add(id(p))
        if out and gap_ok and 0 <= o - out[-1][1] <= gap_elems:
            out[-1][1] = e
            if rode_along is not None:
                rode_along.extend(gap)
        else:
            out.append([o, e])
        gap, gap_ok = [], True
    return [(a, b) for a, b in out]


# MVLT_DDP_NULL_COLLECTIVE=1 (diagnostic, one rank only): the reducer runs -- buckets, events, stream joins, LayerNorm flushes --
# but issues no collective: what is left over a run without a reducer is the reducer's own cost
_NULL_COLLECTIVE = os.environ.get("MVLT_DDP_NULL_COLLECTIVE", "0") == "1"
# MVLT_DDP_DEFER_WAIT=1 (diagnostic, one rank only): the collectives are issued as usual but the end of the backward pass does not
# wait for them -- the optimizer runs beside the last buckets.  (step with the wait) - (step without) = the EXPOSED tail of the
# exchange; (step without the wait) - (step without the collective) = what the collective's kernel costs the backward pass it runs
# beside.  Correct on ONE rank only (an in-place all-reduce over one rank leaves the gradients as they are).
_DEFER_WAIT = os.environ.get("MVLT_DDP_DEFER_WAIT", "0") == "1"
# MVLT_DDP_REHEARSE="blocks[,inflight]" (diagnostic, one rank only): instead of the collective every bucket is copied twice (the
# reduce-scatter and the all-gather pass of a ring both read and write the local buffer once) by a kernel with a ring collective's
# launch geometry -- `blocks` persistent workgroups, a few hundred GB/s -- on a third stream: what the backward pass loses to a
# collective that is xGMI-bound for milliseconds, which RCCL's one-rank memcpy does not show.  Gradients are left unchanged.
_REHEARSE = os.environ.get("MVLT_DDP_REHEARSE", "")


class _EventHandle:
    """Handle of a rehearsal copy: wait() orders the current stream behind it, like a Work handle of the process group."""

    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


class GradReducer:
    def __init__(self, model, bucket_bytes: int = 64 << 20, process_group=None, allow_cpu: bool = False,
                 comm_dtype: torch.dtype = torch.float32, average: bool = True, global_label_mean: bool = True,
                 fork_stream: bool | None = None, merge_gap_elems: int = GAP_ELEMS):
        """average=True (default): p.grad ends up as the MEAN over ranks, like torch DDP, so a stock torch
        optimizer / clip_grad_norm_ on the drop-in sees what it would see on one GPU (RCCL's AVG reduction: no
        extra pass; other backends: SUM + one in-place scale).  average=False leaves the SUM (pair it with
        FusedAdamW(grad_scale=1/world)).
        comm_dtype=torch.bfloat16: buckets are cast to bf16 for the exchange and back afterwards (half the xGMI
        bytes; the cross-rank sum is then rounded to bf16 -- PyTorch DDP's bf16_compress_hook trade-off).  The
        default keeps the reference-equivalent f32 sum.
        fork_stream: issue the buckets from a helper stream that waits for the main AND the weight-gradient stream, so
        the main stream is not stalled at bucket boundaries (one rank over RCCL, B=32: 14.7 ms per step against 15.9 with
        the main stream joining the weight-gradient stream before each bucket; 13.6 without a reducer).  Default (None):
        the MVLT_DDP_FORK environment switch, which is OFF unless set to 1 (see the end of this paragraph).  What orders the exchange: the helper stream waits for everything queued
        on both streams when the bucket leaves (each gradient element is written once per backward pass, before that point
        -- tests/test_model_gpu.py::test_ddp_buckets_carry_final_gradients checks every element, both settings run in
        tests/test_ddp_gpu.py); _finish waits for every handle before the optimizer or the next backward pass touches
        the arena.  NOT yet run over RCCL on more than one GPU by the builder (one-GPU boxes), so the DEFAULT is the
        main-stream join (MVLT_DDP_FORK unset / 0): the collective is then ordered by the stream every gradient kernel was
        queued on or joined into -- correct by construction -- and it is issued one bucket late, behind an event the side
        stream recorded when the bucket closed, so the join does not stall the backward pass (GradReducer._close);
        MVLT_DDP_FORK=1 / fork_stream=True opts into the helper stream."""
        if comm_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("comm_dtype must be float32 or bfloat16")
        import os
        self.comm_dtype = comm_dtype
        self.average = average
        self.use_fork = (os.environ.get("MVLT_DDP_FORK", "0") == "1") if fork_stream is None else bool(fork_stream)
        self.lag = 1              # main-stream mode: buckets are exchanged this many bucket closings late (see _close)
        self.closed = []
        self.global_label_mean = global_label_mean
        self.gap_elems = merge_gap_elems      # plan_ranges: small runs of gradient-less parameters do not split a bucket
        self.comm_buf = None
        self.pending_casts = []
        self.model = model
        self.bucket_elems = bucket_bytes // 4
        # Tapered tail (round 6): what _finish issues at the end of the backward pass is serial behind the
        # last Swin stage-0 kernel at ANY world size -- with one bucket size that is the lagged bucket plus the remainder, up
        # to 2 x 64 MiB.  The gradients that become ready last sit at the LOW end of the arena (patch embedding, Swin stages
        # 0 / 1, the first stage-2 blocks); below `taper_elems` buckets close at `tail_bucket_elems`, so at most two small
        # buckets are still unsent when the backward pass ends.  MVLT_DDP_TAPER_MB / MVLT_DDP_TAIL_BUCKET_MB (0 = no taper).
        taper_mb = float(os.environ.get("MVLT_DDP_TAPER_MB", "32"))
        tail_mb = float(os.environ.get("MVLT_DDP_TAIL_BUCKET_MB", "8"))
        self.taper_elems = int(taper_mb * (1 << 20)) // 4
        self.tail_bucket_elems = min(self.bucket_elems, max(1, int(tail_mb * (1 << 20)) // 4)) if taper_mb > 0 and tail_mb > 0 else self.bucket_elems
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        try:
            self._avg_op = dist.get_backend(process_group) == "nccl"      # ncclAvg; gloo has no AVG
        except Exception:
            self._avg_op = False
        self.allow_cpu = allow_cpu
        self.arena = None
        self.handles = []
        self.launched: List[Tuple[int, int]] = []
        self.done: set = set()
        self.pending_hi = 0
        self.prev_marked = None    # ids of the parameters that received a gradient in the previous completed backward pass
        self.rode_along: list = []  # ids of gradient-less parameters swept into a merged collective in this pass
        self.on_bucket = None      # optional consumer (ranges, handles) of a launched bucket: optim.FusedAdamW overlap
        import sys
        pkg = sys.modules.get(__name__.rsplit(".", 1)[0])
        if getattr(pkg, "HWQ_SET_LATE", False) and not allow_cpu:
            import warnings
            warnings.warn("mvlt_amd.ddp: the HIP runtime was initialised before GPU_MAX_HW_QUEUES could be raised (default 4 hardware "
                          "queues): with RCCL's streams beside the step's two, streams share a queue and serialise (~1 ms per step, "
                          "profiles/r5_ddp_one_rank.md).  Export GPU_MAX_HW_QUEUES=8, or import mvlt_amd before the first CUDA call / "
                          "before init_process_group(device_id=...).", RuntimeWarning, stacklevel=2)
        self.attach()

    def attach(self) -> Arena:
        if self.allow_cpu:
            ar = self.model.__dict__.get("_mvlt_arena") or Arena(self.model, torch.float32, allow_cpu=True)
        else:
            ar = Arena.of(self.model, compute_dtype_of(self.model))
        if ar is not self.arena:
            self.arena = ar
            ar._post_backward = self._finish
            ar._on_watermark = self._on_watermark
            ar._on_backward_begin = self._begin
            dist.broadcast(ar.flat, src=0, group=self.pg)        # identical replicas (bumps flat._version -> bf16 copy refreshed)
        if self.global_label_mean and self.world > 1:
            self.model.__dict__["_mvlt_label_sync"] = self.label_sync
        else:
            self.model.__dict__.pop("_mvlt_label_sync", None)
        return ar

    def no_label_sync(self):
        """Context manager: forwards inside it take no label-count collective (the MLM loss falls back to the per-rank
        mean) -- for a gradient-mode forward that not every rank runs.  NOT torch DDP's no_sync(): gradients of a backward()
        inside it ARE still exchanged (this engine overwrites gradients per backward pass, it does not accumulate them, so
        there is no gradient-accumulation mode to skip the exchange for)."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            had = self.model.__dict__.pop("_mvlt_label_sync", None)
            try:
                yield self
            finally:
                if had is not None:
                    self.model.__dict__["_mvlt_label_sync"] = had
        return ctx()

    def no_sync(self):
        """Deprecated alias of :meth:`no_label_sync` (the name this method had up to round 4; kept so that existing callers and
        loops ported from torch DDP do not break with AttributeError).  NOTE the semantics: gradients of a backward() inside
        the context ARE still exchanged -- only the label-count collective of the forward is skipped."""
        import warnings
        warnings.warn("GradReducer.no_sync() is a deprecated alias of no_label_sync(): gradients are still exchanged inside it "
                      "(this engine overwrites gradients per backward pass; it has no accumulation mode)", DeprecationWarning, stacklevel=2)
        return self.no_label_sync()

    def label_sync(self, count: torch.Tensor) -> torch.Tensor:
        """count: f32 [1] on this rank's device = labelled tokens of this rank's shard.  Returns N_global / world (f32 [1]):
        one 4-byte SUM all-reduce, enqueued on the current stream (no host sync)."""
        g = count.detach().clone()
        err = None
        if g.is_cuda:
            # a hand-off time-out of the fused Swin attention (ops.DeviceHandoffError) poisons THIS rank's loss with NaN,
            # the all-reduce then hands the NaN gradients to every rank -- but only the rank that timed out used to raise, and the
            # others stalled at their next collective.  The sticky error count rides along with the label count (8 bytes instead
            # of 4, same collective): every rank sees the global count and raises one step late (check_handoff, PretrainStep).
            from . import ops
            err = ops.wmsa2_error_tensor(g.device)
        if err is not None:
            g = torch.cat([g, err])
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.pg)
        if err is not None:
            slot = self.__dict__.get("_err_slot")
            if slot is None:
                slot = self._err_slot = [torch.zeros(1, dtype=torch.float32).pin_memory(), None]
            host, ev = slot
            if ev is None or ev.query():          # (never rewrite the pinned word while a copy into it is in flight)
                host.copy_(g[1:2], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                slot[1] = ev
            g = g[:1]
        return g / float(self.world)

    def check_handoff(self) -> None:
        """Raise ops.DeviceHandoffError on EVERY rank when any rank's fused W-MSA hand-off timed out in an earlier step (the
        globally summed count travelled with the label count; looked at without a device sync, so one step late).  The loss of
        that step was NaN and the NaN gradients have been applied by then: the only recovery is the last checkpoint."""
        slot = self.__dict__.get("_err_slot")
        if slot is not None and slot[1] is not None and slot[1].query() and float(slot[0][0]) > 0:
            from . import ops
            n = int(slot[0][0])
            slot[0].zero_()
            ops._wmsa2_raise(n)

    # ---- hooks called by the engines (runtime.backward_begin / arena.watermark / backward_end)
    def _begin(self, arena: Arena) -> None:
        for h in self.__dict__.pop("_deferred", []):
            h.wait()
        self.pending_hi = arena.total
        self.handles, self.launched, self.done = [], [], set()
        self.pending_casts = []
        self.rode_along = []
        self.closed = []          # buckets whose gradients are all queued, not yet exchanged: (lo, hi, side-stream event, ranges)

    def _idle_ok(self, p) -> bool:
        # may this gradient-less parameter's stale slot ride along in a merged collective?  Only when nothing suggests its
        # gradient could still arrive in this pass: frozen, or it had none in the previous completed pass either (the first
        # pass merges nothing)
        return (not p.requires_grad) or (self.prev_marked is not None and id(p) not in self.prev_marked)

    def _on_watermark(self, arena: Arena, lo: int) -> None:
        # bucket size by where the watermark stands: full-size buckets while most of the backward pass is still ahead, small
        # ones for the last `taper_elems` of the arena (scaled down for small arenas so that tests see both regimes)
        taper = min(self.taper_elems, arena.total // 4)
        need = self.tail_bucket_elems if lo < taper else self.bucket_elems
        if self.pending_hi - lo >= need:
            self._close(arena, lo, self.pending_hi)
            self.pending_hi = lo

    def _close(self, arena: Arena, lo: int, hi: int) -> None:
        """Every gradient of [lo, hi) is queued (dgrad chain: main stream; weight gradients: side stream).  Main-stream
        mode (default): the bucket is exchanged ONE bucket later -- the main stream then waits for an event the side stream
        recorded here, which has long passed by the time the next bucket closes, so the backward pass does not stall at
        bucket boundaries the way an immediate join does (one rank, B = 32: 15.9 ms per step with the immediate join, 14.7
        with a helper stream) and the collective is still ordered by the stream every gradient kernel was queued on or
        joined into.  Helper-stream mode (MVLT_DDP_FORK=1) and CPU arenas exchange at once."""
        if not arena.flat.is_cuda or self.use_fork or self.lag <= 0:
            self._launch(arena, lo, hi)
            return
        from . import ops
        ops.LnReduceQueue.flush_all()             # (main stream) LayerNorm gamma / beta gradients of the bucket
        ev = torch.cuda.Event()
        ev.record(ops.side_stream(arena.flat.device))
        # the bucket's CONTENT is fixed at the same instant as the event and the LayerNorm flush: a parameter of [lo, hi) that
        # is marked between now and the (later) launch was not covered by either, so it must not be swept into this
        # collective -- it falls to _finish, which joins the side stream in full
        ranges = plan_ranges(arena, lo, hi, self.done, self.gap_elems, self._idle_ok, self.rode_along)
        self.closed.append((lo, hi, ev, ranges))
        while len(self.closed) > self.lag:
            a, b, e, r = self.closed.pop(0)
            self._launch(arena, a, b, side_event=e, ranges=r)

    def _launch(self, arena: Arena, lo: int, hi: int, side_event=None, ranges=None) -> None:
        fork = None
        if arena.flat.is_cuda:
            from . import ops
            if side_event is None:
                ops.LnReduceQueue.flush_all()     # LayerNorm gamma/beta gradients are reduced in deferred batches
            # The bucket needs the dgrad chain (main stream) AND the weight gradients (side stream).  Joining the
            # side stream into the MAIN stream here would stall the backward pass at every bucket; instead a
            # helper stream waits for both and the collective is issued from it (RCCL's own stream then waits
            # for the helper), so the main stream keeps running ahead.
            if self.use_fork:
                fork = self.__dict__.get("_fork_stream")
                if fork is None:
                    fork = self._fork_stream = torch.cuda.Stream(device=arena.flat.device)
                fork.wait_stream(torch.cuda.current_stream())
                fork.wait_stream(ops.side_stream(arena.flat.device))
        if ranges is None:
            ranges = plan_ranges(arena, lo, hi, self.done, self.gap_elems, self._idle_ok, self.rode_along)
        op = dist.ReduceOp.AVG if (self.average and self._avg_op) else dist.ReduceOp.SUM

        def reduce_ranges(rs):
            if _NULL_COLLECTIVE:          # diagnostic (scripts/r5_ddp_overhead.sh): everything but the collective itself
                return []
            if _REHEARSE and arena.flat.is_cuda and self.world == 1:
                from . import ops
                parts = [int(v) for v in _REHEARSE.split(",")]
                blocks, inflight = parts[0], (parts[1] if len(parts) > 1 else 2)
                st = self.__dict__.get("_rehearse_stream")
                if st is None:
                    st = self._rehearse_stream = torch.cuda.Stream(device=arena.flat.device)
                    self._rehearse_buf = torch.empty(self.bucket_elems + (64 << 10), dtype=torch.float32, device=arena.flat.device)
                st.wait_stream(torch.cuda.current_stream())
                hs = []
                for a, b in rs:
                    for c0 in range(a, b, self._rehearse_buf.numel()):
                        c1 = min(b, c0 + self._rehearse_buf.numel())
                        n = (c1 - c0) // 4 * 4
                        if n <= 0:
                            continue
                        ops.debug_stream_copy(self._rehearse_buf[:n], arena.grad[c0:c0 + n], blocks, inflight, stream=st)
                        ops.debug_stream_copy(arena.grad[c0:c0 + n], self._rehearse_buf[:n], blocks, inflight, stream=st)
                    ev = torch.cuda.Event()
                    ev.record(st)
                    hs.append(_EventHandle(ev))
                return hs
            return [dist.all_reduce(arena.grad[a:b], op=op, group=self.pg, async_op=True) for a, b in rs]

        if self.comm_dtype == torch.float32 and fork is not None and self.on_bucket is None:
            with torch.cuda.stream(fork):
                hs = reduce_ranges(ranges)
        elif self.comm_dtype == torch.float32:
            if arena.flat.is_cuda:
                if side_event is not None:
                    torch.cuda.current_stream().wait_event(side_event)       # recorded when the bucket closed, one bucket ago
                else:
                    torch.cuda.current_stream().wait_stream(ops.side_stream(arena.flat.device))
            hs = reduce_ranges(ranges)
        else:
            if self.on_bucket is not None:
                raise RuntimeError("compressed gradient exchange cannot feed the overlapped optimizer")
            if self.comm_buf is None or self.comm_buf.numel() != arena.total:
                self.comm_buf = torch.empty(arena.total, dtype=self.comm_dtype, device=arena.grad.device)
            hs = []
            if arena.flat.is_cuda:
                if side_event is not None:
                    torch.cuda.current_stream().wait_event(side_event)
                else:
                    torch.cuda.current_stream().wait_stream(ops.side_stream(arena.flat.device))
            for a, b in ranges:
                buf = self.comm_buf[a:b]
                if arena.grad.is_cuda:
                    from . import ops
                    ops.cast(arena.grad[a:b], self.comm_dtype, out=buf)
                else:
                    buf.copy_(arena.grad[a:b])
                hs.append(dist.all_reduce(buf, op=op, group=self.pg, async_op=True))
                self.pending_casts.append((a, b))
        self.handles += hs
        self.launched += ranges
        if self.on_bucket is not None and ranges:
            # a backend without an AVG reduction (gloo) is rescaled in _finish, AFTER this consumer has used the
            # bucket: tell it the factor that is still owed
            owed = 1.0 / self.world if (self.average and not self._avg_op and self.world > 1) else 1.0
            self.on_bucket(arena, ranges, hs, owed)

    def _finish(self, arena: Arena) -> None:
        # everything not yet communicated, including parameters whose gradient
        # arrived out of watermark order (independent head branches)
        late = [pid for pid in self.rode_along if arena.has_grad[pid]]
        if late:
            # a parameter that rode along in a merged collective as "gradient-less" was written afterwards: its slot was
            # reduced in place while (or before) its gradient arrived, and would be reduced again below
            raise RuntimeError(f"mvlt_amd.ddp: {len(late)} parameter(s) received a gradient after their arena slot had left in "
                               "a merged bucket; construct GradReducer(merge_gap_elems=0) for models with data-dependent branches")
        for a, b, e, r in self.closed:           # buckets still waiting for their turn (main-stream mode)
            self._launch(arena, a, b, side_event=e, ranges=r)
        self.closed = []
        self._launch(arena, 0, arena.total)
        self.pending_hi = 0
        self.prev_marked = {id(p) for p in arena._marked}
        if _DEFER_WAIT and self.world == 1 and arena.flat.is_cuda:
            self._deferred = self.handles          # diagnostic: waited for at the start of the next backward pass
        else:
            for h in self.handles:
                h.wait()
        self.handles = []
        if arena.flat.is_cuda:
            from . import ops
            ops.join_side(arena.flat.device)      # (also releases the tensors kept alive for the side stream)
        for a, b in self.pending_casts:          # compressed exchange: summed bf16 -> f32 gradient arena
            if arena.grad.is_cuda:
                from . import ops
                ops.cast(self.comm_buf[a:b], torch.float32, out=arena.grad[a:b])
            else:
                arena.grad[a:b].copy_(self.comm_buf[a:b])
        self.pending_casts = []
        if self.average and not self._avg_op and self.world > 1:      # no AVG reduction on this backend (gloo)
            for a, b in self.launched:
                arena.grad[a:b].mul_(1.0 / self.world)
