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
