"""Property tests (hypothesis) of the integer tables and the sharding logic: size-independent invariants."""
import os
import sys

import torch
from hypothesis import given, settings, strategies as st

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mvlt_amd import indexing as I            # noqa: E402
from mvlt_amd.data import ShardSampler          # noqa: E402


@settings(max_examples=40, deadline=None)
@given(st.integers(1, 8), st.integers(1, 8), st.sampled_from([0, 1, 3]), st.integers(1, 3))
def test_window_maps_are_inverse_permutations(nh, nw, shift, B):
    """roll(-shift) + window_partition is a permutation of the tokens; the batched maps invert each other and the
    map equals torch.roll + the reference's view/permute chain (visual_feature_extractor.py:144-156, :360-367)."""
    ws = 7
    H, W = nh * ws, nw * ws
    src = I.window_token_map(H, W, ws, shift)
    assert sorted(src.tolist()) == list(range(H * W))
    x = torch.arange(H * W).view(1, H, W, 1)
    rolled = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2)) if shift else x
    ref = rolled.view(1, H // ws, ws, W // ws, ws, 1).permute(0, 1, 3, 2, 4, 5).reshape(-1)
    assert torch.equal(src, ref)
    w2n, n2w = I.batched_window_maps(B, H, W, ws, shift, "cpu")
    assert torch.equal(w2n[n2w.long()].long(), torch.arange(B * H * W))
    assert torch.equal(n2w[w2n.long()].long(), torch.arange(B * H * W))
    assert int(w2n.max()) == B * H * W - 1 and (w2n.view(B, -1) // (H * W) == torch.arange(B)[:, None]).all()


@settings(max_examples=30, deadline=None)
@given(st.integers(1, 10), st.integers(1, 10))
def test_patch_merge_map_is_a_partition(h2, w2):
    H, W = 2 * h2, 2 * w2
    m = I.patch_merge_map(H, W)
    assert m.shape == (h2 * w2, 4) and sorted(m.reshape(-1).tolist()) == list(range(H * W))
    x = torch.arange(H * W).view(1, H, W, 1)           # visual_feature_extractor.py:435-439
    ref = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1).view(-1, 4)
    assert torch.equal(m, ref)


@settings(max_examples=30, deadline=None)
@given(st.integers(1, 6), st.integers(1, 6), st.sampled_from([1, 3]))
def test_shift_mask_is_symmetric_zero_or_minus_100_and_empty_inside(nh, nw, shift):
    ws = 7
    H, W = nh * ws, nw * ws
    m = I.shift_attn_mask(H, W, ws, shift)
    assert m.shape == (nh * nw, 49, 49) and set(m.unique().tolist()) <= {0.0, -100.0}
    assert torch.equal(m, m.transpose(1, 2)) and (m.diagonal(dim1=1, dim2=2) == 0).all()
    interior = [w for w in range(nh * nw) if w // nw != nh - 1 and w % nw != nw - 1]
    assert all((m[w] == 0).all() for w in interior)    # only the last window row / column straddles a border


@settings(max_examples=60, deadline=None)
@given(st.integers(1, 200), st.integers(1, 8), st.booleans(), st.booleans(), st.integers(0, 5))
def test_shard_sampler_partitions_the_epoch(n, world, shuffle, drop_last, epoch):
    per_rank = []
    for r in range(world):
        s = ShardSampler(n, world, r, shuffle=shuffle, seed=3, drop_last=drop_last)
        s.set_epoch(epoch)
        idx = list(s)
        assert len(idx) == len(s)
        per_rank.append(idx)
    assert len({len(p) for p in per_rank}) == 1                       # same number of samples on every rank
    allidx = [i for p in per_rank for i in p]
    assert all(0 <= i < n for i in allidx)
    if drop_last and n % world:
        if n >= world:
            assert len(set(allidx)) == len(allidx)                    # nothing repeated when the tail is dropped
    else:
        assert set(allidx) == set(range(n))                           # every sample seen (tail padded by repetition)
