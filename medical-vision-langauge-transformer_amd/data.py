"""Input pipeline for pre-training on pre-resized / pre-tokenised shards.

The reference builds every sample on the host inside ``Dataset.__getitem__``
(``run_pretrain_rgc_roco_medicat.py:94-212``: PIL decode + resize, per-channel
``(x - mean) / var``, WordPiece tokenisation, ``_random_mask_word``, ITM negative
sampling) with 8 DataLoader workers.  At ~1.9 k pairs/s per GPU that host work is
the bottleneck, so the per-step arithmetic moves to the GPU and the per-sample
decode/tokenise work moves offline:

* ``write_shard`` / ``Shard``: uint8 HWC images already resized to 224x224, token
  ids already truncated with the reference rule (``ids[:T-1] + [END]``, :170-172)
  and the untruncated token count, as three ``.npy`` files (memory-mapped);
* ``itm_pairs``: the ITM negative sampling of ``__getitem__`` (:134-158) on indices;
* ``ShardSampler``: the index arithmetic of ``torch.utils.data.DistributedSampler``
  (one process per GPU), bit-identical to it;
* ``normalize_images`` / ``mask_captions``: ``mvlt_image_normalize`` / ``mvlt_mlm_mask``;
* ``PretrainBatches``: ties them together and yields the 5-tuple ``PretrainStep`` takes
  (the fifth element, the caption lengths, enables packed rows).
"""
from __future__ import annotations

import ctypes as C
import os
import random
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L
from . import ops


# ----------------------------------------------------------------------------- GPU steps
def normalize_images(u8_hwc: torch.Tensor) -> torch.Tensor:
    """uint8 [B,H,W,3] (RGB, as PIL gives it) -> f32 [B,3,H,W], per image and channel (x - mean) / var
    (run_pretrain_rgc_roco_medicat.py:107-110 -- np.var, i.e. the variance, as the reference has it)."""
    if not u8_hwc.is_cuda:
        raise RuntimeError("mvlt_amd runs on the GPU only (no CPU fallback)")
    assert u8_hwc.dtype == torch.uint8 and u8_hwc.dim() == 4 and u8_hwc.shape[3] == 3 and u8_hwc.is_contiguous()
    B, H, W, _ = u8_hwc.shape
    out = torch.empty((B, 3, H, W), dtype=torch.float32, device=u8_hwc.device)
    L.check(L.lib().mvlt_image_normalize(ops._p(u8_hwc), ops._p(out), B, H, W, ops._stream()), "mvlt_image_normalize")
    return out


def mask_captions(ids: torch.Tensor, full_len: torch.Tensor, seed: int, vocab_size: int, mask_id: int = 103,
                  itm_label: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """``_random_mask_word`` (:188-212) for a batch of truncated id rows -> (caption_masked, caption_label)."""
    if not ids.is_cuda:
        raise RuntimeError("mvlt_amd runs on the GPU only (no CPU fallback)")
    assert ids.dtype == torch.int64 and ids.dim() == 2 and ids.is_contiguous()
    assert full_len.dtype == torch.int32 and full_len.numel() == ids.shape[0] and full_len.is_cuda
    out, labels = torch.empty_like(ids), torch.empty_like(ids)
    p = L.MvltMlmMask()
    p.B, p.T, p.vocab_size, p.mask_id = ids.shape[0], ids.shape[1], int(vocab_size), int(mask_id)
    p.ids_in, p.full_len, p.ids_out, p.labels = ops._p(ids), ops._p(full_len), ops._p(out), ops._p(labels)
    if itm_label is not None:
        assert itm_label.dtype == torch.int64 and itm_label.numel() == ids.shape[0]
        p.itm_label = ops._p(itm_label)
    p.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    L.check(L.lib().mvlt_mlm_mask(C.byref(p), ops._stream()), "mvlt_mlm_mask")
    return out, labels


# ----------------------------------------------------------------------------- host logic (pure integer work)
def truncate_ids(token_ids: Sequence[int], T: int) -> Tuple[np.ndarray, int]:
    """:166-176 -- keep the first T-1 ids and the last one ([END]); zero-pad to T.  Returns (row, full length)."""
    n = len(token_ids)
    ids = list(token_ids)
    if n > T:
        ids = ids[:T - 1] + [ids[-1]]
    row = np.zeros(T, dtype=np.int64)
    row[:len(ids)] = ids
    return row, n


def deal_balanced(items: Sequence[int], weights: Sequence[int], parts: int) -> List[List[int]]:
    """Deal ``items`` (len = parts * k) into ``parts`` lists of k items with nearly equal weight sums: heaviest first, each
    to the lightest list that still has room (longest-processing-time rule; deterministic, ties by list index).  With the
    encoder running on packed rows a rank's step time follows the SUM of its caption lengths: an unbalanced deal makes
    every rank wait at the gradient all-reduce for the one with the longest captions (8 ranks x 32 samples of U{16..79}
    tokens: the slowest rank carries ~+4.7 % rows; dealt this way the sums differ by less than one caption)."""
    k = len(items) // parts
    assert k * parts == len(items) == len(weights)
    order = sorted(range(len(items)), key=lambda i: (-int(weights[i]), i))
    out: List[List[int]] = [[] for _ in range(parts)]
    load = [0] * parts
    for i in order:
        r = min((r for r in range(parts) if len(out[r]) < k), key=lambda r: (load[r], r))
        out[r].append(items[i])
        load[r] += int(weights[i])
    return out


class ShardSampler:
    """Indices of ``torch.utils.data.DistributedSampler(dataset, num_replicas, rank, shuffle, seed, drop_last)``:
    same permutation (torch.Generator seeded with seed + epoch), same padding, same rank stride.
    ``lengths`` (caption length per dataset index) + ``batch_size``: every GLOBAL batch (the num_replicas * batch_size
    entries the ranks would draw for one step) keeps its members but is re-dealt to the ranks with ``deal_balanced`` -- the
    global-batch gradient is unchanged, the ranks' packed row counts are equalised."""

    def __init__(self, n: int, num_replicas: int = 1, rank: int = 0, shuffle: bool = True, seed: int = 0,
                 drop_last: bool = False, lengths: Optional[Sequence[int]] = None, batch_size: Optional[int] = None):
        if not 0 <= rank < num_replicas:
            raise ValueError("rank out of range")
        self.n, self.num_replicas, self.rank = n, num_replicas, rank
        self.shuffle, self.seed, self.drop_last, self.epoch = shuffle, seed, drop_last, 0
        if drop_last and n % num_replicas != 0:
            self.num_samples = -(-(n - num_replicas) // num_replicas)
        else:
            self.num_samples = -(-n // num_replicas)
        self.total_size = self.num_samples * num_replicas
        if (lengths is None) != (batch_size is None):
            raise ValueError("length balancing needs both lengths and batch_size")
        self.lengths, self.batch_size = lengths, batch_size

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def __len__(self) -> int:
        return self.num_samples

    def __iter__(self) -> Iterator[int]:
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            idx = torch.randperm(self.n, generator=g).tolist()
        else:
            idx = list(range(self.n))
        if not self.drop_last:
            pad = self.total_size - len(idx)
            if pad > 0:
                idx += (idx * (-(-pad // len(idx))))[:pad]
        else:
            idx = idx[:self.total_size]
        if self.lengths is None or self.num_replicas == 1:
            return iter(idx[self.rank:self.total_size:self.num_replicas])
        mine: List[int] = []
        step = self.num_replicas * self.batch_size
        for s0 in range(0, self.total_size, step):
            chunk = idx[s0:s0 + step]
            if len(chunk) < step:          # ragged tail: the plain stride
                mine += chunk[self.rank::self.num_replicas]
            else:
                mine += deal_balanced(chunk, [self.lengths[i] for i in chunk], self.num_replicas)[self.rank]
        return iter(mine)


def itm_pairs(indices: Sequence[int], n_total: int, cap_id_of, rng: random.Random, itm_task: bool = True
              ) -> List[Tuple[int, int, int]]:
    """(image index, caption index, ITM label) per sample, following ``__getitem__`` (:134-158): with
    probability 1/2 the pair is kept (label 1); otherwise a random other sample with a different caption id
    replaces either the image or the caption (probability 1/2 each) and the label is 0."""
    out = []
    for i in indices:
        if rng.random() < 0.5 or not itm_task:
            out.append((i, i, 1))
            continue
        j = rng.randrange(0, n_total)
        while j == i or cap_id_of(j) == cap_id_of(i):
            j = rng.randrange(0, n_total)
        out.append((j, i, 0) if rng.random() < 0.5 else (i, j, 0))
    return out


# ----------------------------------------------------------------------------- shards
def write_shard(path: str, images_u8_hwc: np.ndarray, id_rows: np.ndarray, full_len: np.ndarray) -> None:
    os.makedirs(path, exist_ok=True)
    assert images_u8_hwc.dtype == np.uint8 and images_u8_hwc.ndim == 4 and images_u8_hwc.shape[3] == 3
    assert id_rows.ndim == 2 and len(id_rows) == len(images_u8_hwc) == len(full_len)
    np.save(os.path.join(path, "images_u8.npy"), images_u8_hwc)
    np.save(os.path.join(path, "ids.npy"), id_rows.astype(np.int64))
    np.save(os.path.join(path, "full_len.npy"), full_len.astype(np.int32))


class Shard:
    def __init__(self, path: str):
        self.images = np.load(os.path.join(path, "images_u8.npy"), mmap_mode="r")
        self.ids = np.load(os.path.join(path, "ids.npy"), mmap_mode="r")
        self.full_len = np.load(os.path.join(path, "full_len.npy"), mmap_mode="r")

    def __len__(self) -> int:
        return len(self.ids)


class PretrainBatches:
    """Iterates (image, caption_masked, caption_label, image_text_label, text_lengths) batches: gathers the uint8
    images / id rows of a batch on the host (pinned), one async H2D copy each, normalisation + masking on the GPU.
    ``text_lengths`` stays on the host (it is what ``PretrainStep`` needs for packed rows)."""

    def __init__(self, shard: Shard, batch_size: int, device, num_replicas: int = 1, rank: int = 0, seed: int = 0,
                 vocab_size: int = 30522, mask_id: int = 103, itm_task: bool = True, mlm_task: bool = True,
                 drop_last: bool = True):
        self.shard, self.B, self.device = shard, batch_size, torch.device(device)
        self.sampler = ShardSampler(len(shard), num_replicas, rank, shuffle=True, seed=seed, drop_last=drop_last)
        self.rng = random.Random(seed * 7919 + rank)
        self.seed, self.vocab_size, self.mask_id, self.itm_task, self.mlm_task = seed, vocab_size, mask_id, itm_task, mlm_task
        self.step = 0

    def set_epoch(self, epoch: int) -> None:
        self.sampler.set_epoch(epoch)

    def __iter__(self):
        idx = list(self.sampler)
        T = self.shard.ids.shape[1]
        for s in range(0, len(idx) - self.B + 1, self.B):
            pairs = itm_pairs(idx[s:s + self.B], len(self.shard), lambda k: k, self.rng, self.itm_task)
            img_i = np.fromiter((p[0] for p in pairs), dtype=np.int64)
            cap_i = np.fromiter((p[1] for p in pairs), dtype=np.int64)
            img = torch.from_numpy(np.ascontiguousarray(self.shard.images[img_i])).pin_memory()
            ids = torch.from_numpy(np.ascontiguousarray(self.shard.ids[cap_i])).pin_memory()
            flen = torch.from_numpy(np.ascontiguousarray(self.shard.full_len[cap_i]))
            itm = torch.tensor([p[2] for p in pairs], dtype=torch.int64)
            image = normalize_images(img.to(self.device, non_blocking=True))
            ids_d = ids.to(self.device, non_blocking=True)
            itm_d = itm.pin_memory().to(self.device, non_blocking=True)
            if self.mlm_task:
                masked, labels = mask_captions(ids_d, flen.pin_memory().to(self.device, non_blocking=True),
                                               seed=self.seed * 1000003 + self.step, vocab_size=self.vocab_size,
                                               mask_id=self.mask_id, itm_label=itm_d)
            else:
                masked, labels = ids_d, torch.full_like(ids_d, -100)
            self.step += 1
            yield image, masked, labels, itm_d, torch.clamp(flen, max=T).to(torch.int32)
