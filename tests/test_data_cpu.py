"""Host logic of the input pipeline (mvlt_amd/data.py): integer work, bit-exact."""
import os
import random
import sys

import pytest
import torch
from torch.utils.data import DistributedSampler

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


@pytest.mark.parametrize("n,world", [(10, 1), (10, 3), (17, 4), (64, 8), (5, 8)])
@pytest.mark.parametrize("shuffle", [True, False])
@pytest.mark.parametrize("drop_last", [False, True])
def test_shard_sampler_equals_torch_distributed_sampler(n, world, shuffle, drop_last):
    from mvlt_amd.data import ShardSampler
    if drop_last and n < world:
        pytest.skip("torch's DistributedSampler yields nothing meaningful here")
    ds = list(range(n))
    for rank in range(world):
        for epoch in (0, 3):
            ref = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=shuffle, seed=11, drop_last=drop_last)
            ref.set_epoch(epoch)
            mine = ShardSampler(n, world, rank, shuffle=shuffle, seed=11, drop_last=drop_last)
            mine.set_epoch(epoch)
            assert list(mine) == list(ref) and len(mine) == len(ref)


def test_truncate_ids_reference_rule():
    """run_pretrain_rgc_roco_medicat.py:166-176: first T-1 ids + the last one, zero padded."""
    from mvlt_amd.data import truncate_ids
    row, n = truncate_ids([5, 6, 7, 104], 8)
    assert row.tolist() == [5, 6, 7, 104, 0, 0, 0, 0] and n == 4
    row, n = truncate_ids(list(range(1, 13)) + [104], 8)
    assert row.tolist() == [1, 2, 3, 4, 5, 6, 7, 104] and n == 13
    row, n = truncate_ids(list(range(1, 8)) + [104], 8)
    assert row.tolist() == [1, 2, 3, 4, 5, 6, 7, 104] and n == 8


def test_itm_pairs_follow_getitem():
    from mvlt_amd.data import itm_pairs
    rng = random.Random(3)
    cap = lambda k: k // 2          # two images share a caption id
    pairs = itm_pairs(list(range(400)), 400, cap, rng)
    pos = [p for p in pairs if p[2] == 1]
    neg = [p for p in pairs if p[2] == 0]
    assert 150 < len(pos) < 250 and len(pos) + len(neg) == 400
    assert all(i == j for i, j, _ in pos)
    swapped_img = sum(1 for (i, j, _), k in zip(pairs, range(400)) if _ == 0 and j == k)
    assert 0.3 * len(neg) < swapped_img < 0.7 * len(neg)
    for (i, j, l), k in zip(pairs, range(400)):
        if l == 0:
            other = i if j == k else j
            assert other != k and cap(other) != cap(k) and (i == k or j == k)
    assert all(l == 1 for _, _, l in itm_pairs(list(range(50)), 50, cap, rng, itm_task=False))


def test_length_balanced_sharding_keeps_the_global_batch():
    """ShardSampler(lengths=, batch_size=): every global step draws the same members as DistributedSampler's stride, re-dealt so
    that the ranks' caption-length sums (their packed encoder rows) differ by less than one caption."""
    import random
    from mvlt_amd.data import ShardSampler, deal_balanced
    n, world, bs = 1000, 8, 32
    rng = random.Random(3)
    lengths = [rng.randint(16, 79) for _ in range(n)]
    plain = [list(ShardSampler(n, world, r, seed=5)) for r in range(world)]
    bal = [list(ShardSampler(n, world, r, seed=5, lengths=lengths, batch_size=bs)) for r in range(world)]
    assert all(len(b) == len(p) for b, p in zip(bal, plain))
    steps = len(plain[0]) // bs
    worst_plain = worst_bal = 0
    for s in range(steps):
        members_plain = sorted(i for r in range(world) for i in plain[r][s * bs:(s + 1) * bs])
        members_bal = sorted(i for r in range(world) for i in bal[r][s * bs:(s + 1) * bs])
        assert members_plain == members_bal                      # same global batch
        sums_p = [sum(lengths[i] for i in plain[r][s * bs:(s + 1) * bs]) for r in range(world)]
        sums_b = [sum(lengths[i] for i in bal[r][s * bs:(s + 1) * bs]) for r in range(world)]
        worst_plain = max(worst_plain, max(sums_p) - min(sums_p))
        worst_bal = max(worst_bal, max(sums_b) - min(sums_b))
    assert worst_bal <= 79 and worst_bal < worst_plain / 4, (worst_bal, worst_plain)
    parts = deal_balanced(list(range(16)), [5, 1, 9, 3, 7, 7, 2, 8, 4, 6, 1, 1, 9, 9, 3, 5], 4)
    assert sorted(i for p in parts for i in p) == list(range(16)) and all(len(p) == 4 for p in parts)
