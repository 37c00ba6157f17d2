"""Small per-model runtime state: compute dtype and dropout seed stream."""
import torch
import torch.nn as nn

DEFAULT_COMPUTE_DTYPE = torch.bfloat16
_seed_state = {"base": 0x5EED1234, "counter": 0}


def set_compute_dtype(model: nn.Module, dtype: torch.dtype) -> nn.Module:
    """torch.bfloat16 (MFMA bf16, f32 accumulate; default) or torch.float32
    (exact f32 MFMA -- the reference's own precision, used for tight parity)."""
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("compute dtype must be float32 or bfloat16")
    for m in model.modules():
        m.__dict__["_mvlt_compute_dtype"] = dtype
    return model


def compute_dtype_of(module: nn.Module) -> torch.dtype:
    return module.__dict__.get("_mvlt_compute_dtype", DEFAULT_COMPUTE_DTYPE)


def manual_seed(seed: int) -> None:
    """Seed of the counter-based dropout RNG (csrc/common.h rng_u32)."""
    _seed_state["base"] = int(seed) & 0xFFFFFFFFFFFF
    _seed_state["counter"] = 0


def next_seed() -> int:
    """One fresh 64-bit seed per forward pass; (seed, tag, index) -> mask."""
    _seed_state["counter"] += 1
    return (_seed_state["base"] * 1000003 + _seed_state["counter"] * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF


def backward_begin(arena) -> None:
    """Called by every engine backward: the first one of a pass resets the
    gradient bookkeeping and queues the end-of-backward callback."""
    if not arena.__dict__.get("_in_backward", False):
        arena._in_backward = True
        arena.begin_backward()
        torch.autograd.Variable._execution_engine.queue_callback(lambda: backward_end(arena))


def backward_end(arena) -> None:
    arena._in_backward = False
    hook = arena.__dict__.get("_post_backward")
    if hook is not None:
        hook(arena)
    arena.publish_grads()


# ----------------------------------------------------------------------------------------------- replayed inference forward
class GraphedEval:
    """An inference forward (eval mode, no autograd) captured ONCE per input signature as a HIP graph and replayed.

    Why: a forward pass is ~330 launches of 2-10 us each; at the batch sizes the reference's Med-VQA script uses (B = 2,
    run_vqa.py / BASELINE config #1) the host cannot enqueue them as fast as the GPU runs them (3.7 ms per call against
    ~1.5 ms of device time).  A replay costs one host call.  The captured sequence is the very same launch sequence (same
    kernels, same order, the side-stream fork / join of the packing plan included), so results are bit-identical to the eager call.

    ``fn(*tensors)`` must be a pure function of its tensor arguments' VALUES for fixed shapes / dtypes: no host-side decisions
    that depend on them, no dropout (eval), parameters read through the arena (their storage never moves; the bf16 compute
    copy is refreshed before every replay).  Arguments may be tensors or None.  Outputs are returned as fresh clones.
    The first ``warmup`` calls of a signature run eagerly (lazy state, allocator); capture happens on a private stream."""

    MAX_GRAPHS = 8       # signatures kept captured per module (least recently used beyond that is dropped; ADVICE r5: variable
                         # question lengths / batch sizes must not grow graphs and their memory without bound)

    def __init__(self, fn, module, warmup=2):
        self.fn, self.module, self.warmup = fn, module, warmup
        self.seen, self.graphs = {}, {}
        self._pool = None    # one private memory pool shared by every graph of this module

    @staticmethod
    def _sig(args):
        return tuple(None if a is None else (tuple(a.shape), a.dtype, a.device.index) for a in args)

    def __call__(self, *args):
        from . import ops
        from .arena import Arena
        if torch.is_grad_enabled() or self.module.training or any(a is not None and not a.is_cuda for a in args):
            return self.fn(*args)
        ar = Arena.of(self.module, compute_dtype_of(self.module))
        key = (self._sig(args), ar.flat.data_ptr(), compute_dtype_of(self.module))
        ent = self.graphs.get(key)
        if ent is None:
            n = self.seen.get(key, 0)
            self.seen[key] = n + 1
            if n < self.warmup:
                return self.fn(*args)
            if len(self.graphs) >= self.MAX_GRAPHS:
                del self.graphs[next(iter(self.graphs))]          # dicts keep insertion order: the least recently used entry
                if len(self.seen) > 64:
                    self.seen.clear()
            ent = self.graphs[key] = self._capture(args)
        else:
            self.graphs[key] = self.graphs.pop(key)              # most recently used last
        static_in, static_out, graph = ent
        for s, a in zip(static_in, args):
            if s is not None:
                s.copy_(a, non_blocking=True)
        ar.refresh_shadow()                      # parameters written since the last call (load_state_dict, an optimizer step)
        graph.replay()
        # a hand-off wait of the fused Swin attention that ran out inside the replay poisons the outputs with NaN; the sticky
        # count is looked at here without a sync (raises one call late, like PretrainStep) -- ADVICE r5
        ops.wmsa2_check(sync=False)
        outs = tuple(o.clone() if isinstance(o, torch.Tensor) else o for o in static_out)
        return outs if len(outs) != 1 else outs[0]

    def _capture(self, args):
        from . import ops
        static_in = [None if a is None else a.clone() for a in args]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), ops.on_stream(side, "graph"):      # once more on the capture stream: per-stream state
            self.fn(*static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        ops.pin_scratch()                        # workspaces whose addresses the graph records must never be freed
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=self._pool):
            with ops.on_stream(torch.cuda.current_stream(), "graph"):
                out = self.fn(*static_in)
        if self._pool is None:
            self._pool = g.pool()
        ops.pin_scratch()
        out = out if isinstance(out, tuple) else (out,)
        return static_in, out, g
