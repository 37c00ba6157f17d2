"""world_size-2 gloo test (CPU) of the data-parallel gradient exchange:
bucket planning over the flat arena, watermark-driven launches, parameters
without a gradient are not communicated, replicas are broadcast from rank 0."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(300, 200)      # "early" layer (low offsets)
        self.b = nn.Linear(200, 100)
        self.idle = nn.Linear(100, 50)    # never gets a gradient
        self.c = nn.Linear(100, 10)       # "late" layer (high offsets)


def _worker(rank, world, port, q, bf16_comm=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from mvlt_amd.arena import Arena
        from mvlt_amd.ddp import GradReducer
        from mvlt_amd import runtime
        torch.manual_seed(rank)             # different init per rank -> broadcast must equalise
        model = Toy()
        red = GradReducer(model, bucket_bytes=64 * 1024, allow_cpu=True,
                          comm_dtype=torch.bfloat16 if bf16_comm else torch.float32)
        ar = red.arena
        ref0 = [torch.zeros_like(ar.flat) for _ in range(world)]
        dist.all_gather(ref0, ar.flat)
        assert all(torch.equal(ref0[0], t) for t in ref0), "replicas differ after broadcast"
        # emulate one backward pass: engines mark parameters from the end of the arena downwards
        runtime_calls = []
        ar.begin_backward()
        g = torch.Generator().manual_seed(100 + rank)
        local = {}
        for mod in (model.c, model.b, model.a):                 # descending offsets
            for p in (mod.weight, mod.bias):
                v = ar.grad_view(p)
                v.copy_(torch.randn(v.shape, generator=g))
                local[id(p)] = v.clone()
            ar.mark(mod.weight, mod.bias)
        assert len(red.launched) >= 1, "watermark should have launched at least one bucket before the end"
        early = list(red.launched)
        runtime.backward_end(ar)
        # every active parameter reduced exactly once, idle one untouched
        covered = sorted(red.launched)
        for (a0, b0), (a1, b1) in zip(covered, covered[1:]):
            assert b0 <= a1, "overlapping all-reduce ranges"
        idle_o = ar.offset[id(model.idle.weight)]
        assert not any(a <= idle_o < b for a, b in covered), "idle parameter was communicated"
        assert model.idle.weight.grad is None and model.a.weight.grad is not None
        # sum over ranks == sum of the per-rank gradients
        for mod in (model.a, model.b, model.c):
            for p in (mod.weight, mod.bias):
                mine = [torch.zeros_like(local[id(p)]) for _ in range(world)]
                dist.all_gather(mine, local[id(p)])
                if bf16_comm:      # each rank's gradient and the sum are rounded to bf16 on the wire
                    assert torch.allclose(ar.grad_view(p), sum(mine), rtol=2e-2, atol=2e-2)
                else:
                    assert torch.allclose(ar.grad_view(p), sum(mine), atol=1e-6)
        q.put((rank, "ok", len(early), len(covered)))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "fail: " + traceback.format_exc(), 0, 0))
    finally:
        dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("bf16_comm", [False, True])
def test_gradient_exchange_two_ranks_gloo(bf16_comm):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, bf16_comm)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(r[1] == "ok" for r in res), res


def test_coin_flip_is_shared():
    """Ranks seeded with seed_coin_flip draw the same seq2seq/bidir sequence (model.py:390)."""
    import random
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mvlt_amd.ddp import seed_coin_flip
    seqs = []
    for _rank in range(2):
        seed_coin_flip(5678)
        seqs.append([random.random() < 0.5 for _ in range(32)])
    assert seqs[0] == seqs[1] and 4 < sum(seqs[0]) < 28
