"""Input-pipeline kernels (mvlt_image_normalize, mvlt_mlm_mask) against the reference's host arithmetic
(run_pretrain_rgc_roco_medicat.py:107-110, :188-212) and the batch iterator end to end."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_image_normalize_matches_the_numpy_statement():
    from mvlt_amd.data import normalize_images
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(3, 224, 224, 3), dtype=np.uint8)
    img[1, :, :, 2] = (rng.integers(0, 4, size=(224, 224)) * 60).astype(np.uint8)     # low-variance channel
    out = normalize_images(torch.from_numpy(img).cuda()).cpu().numpy()
    for b in range(3):
        im_np = np.transpose(np.array(img[b], dtype=np.float32), (2, 0, 1))          # the reference's own lines
        for c in range(3):
            im_np[c] = (im_np[c] - np.mean(im_np[c])) / np.var(im_np[c])
        assert out[b].shape == im_np.shape
        assert np.abs(out[b] - im_np).max() <= 2e-5 * np.abs(im_np).max()
    const = np.full((1, 8, 8, 3), 7, dtype=np.uint8)
    assert torch.isnan(normalize_images(torch.from_numpy(const).cuda())).all()        # 0/0, as numpy gives


def test_mlm_mask_follows_random_mask_word():
    from mvlt_amd.data import mask_captions, truncate_ids
    T, B, V = 24, 4096, 3000
    g = torch.Generator().manual_seed(1)
    rows, full = [], []
    for b in range(B):
        n = int(torch.randint(3, 40, (1,), generator=g))            # some captions are longer than T
        toks = torch.randint(1000, V, (n,), generator=g).tolist()
        toks[-1] = 104
        r, f = truncate_ids(toks, T)
        rows.append(r); full.append(f)
    ids = torch.from_numpy(np.stack(rows)).cuda()
    flen = torch.tensor(full, dtype=torch.int32).cuda()
    itm = (torch.arange(B) % 5 != 0).long().cuda()                   # every 5th row is an ITM negative
    out, lab = mask_captions(ids, flen, seed=77, vocab_size=V, mask_id=103, itm_label=itm)
    out2, lab2 = mask_captions(ids, flen, seed=77, vocab_size=V, mask_id=103, itm_label=itm)
    assert torch.equal(out, out2) and torch.equal(lab, lab2)         # pure function of (seed, row, position)
    out3, _ = mask_captions(ids, flen, seed=78, vocab_size=V, mask_id=103, itm_label=itm)
    assert not torch.equal(out, out3)
    ids_c, out_c, lab_c = ids.cpu(), out.cpu(), lab.cpu()
    picked = lab_c >= 0
    assert not picked[::5].any() and torch.equal(out_c[::5], ids_c[::5])        # ITM negatives: untouched
    assert torch.equal(lab_c[picked], ids_c[picked])                            # label = original id
    assert torch.equal(out_c[~picked], ids_c[~picked])                          # nothing else changes
    assert not (picked & (ids_c == 0)).any()                                    # never a padding column
    n_kept, n_mask, n_rand, n_same = 0, 0, 0, 0
    for b in range(B):
        if b % 5 == 0:
            continue
        want = min(10, max(1, round(full[b] * 0.2)))                            # Python round(), the reference's
        got = int(picked[b].sum())
        if full[b] <= T:
            assert got == want, (b, full[b], got, want)
        else:
            assert got <= want                                                  # masks beyond the cut are dropped
        n_kept += got
    ch = picked & (out_c != ids_c)
    n_mask = int((ch & (out_c == 103)).sum()); n_rand = int((ch & (out_c != 103)).sum())
    n_same = int((picked & (out_c == ids_c)).sum())
    assert abs(n_mask / n_kept - 0.8) < 0.02 and abs(n_rand / n_kept - 0.1) < 0.015 and abs(n_same / n_kept - 0.1) < 0.015
    assert int(out_c[ch & (out_c != 103)].max()) < V
    # uniform choice of positions: first and last real position are picked equally often (+-)
    short = [b for b in range(B) if b % 5 and 10 <= full[b] <= T]
    first = sum(int(picked[b, 0]) for b in short) / len(short)
    last = sum(int(picked[b, full[b] - 1]) for b in short) / len(short)
    assert abs(first - last) < 0.05 and 0.1 < first < 0.3


def test_pretrain_batches_feed_a_training_step(tmp_path):
    import mvlt_amd as M
    from mvlt_amd.data import PretrainBatches, Shard, truncate_ids, write_shard
    from mvlt_amd.train import PretrainStep
    rng = np.random.default_rng(3)
    N, T, V = 12, 24, 3000
    images = rng.integers(0, 256, size=(N, 224, 224, 3), dtype=np.uint8)
    rows, full = zip(*[truncate_ids(list(rng.integers(1000, V, size=int(rng.integers(4, 30)))) + [104], T) for _ in range(N)])
    write_shard(str(tmp_path / "s0"), images, np.stack(rows), np.array(full))
    cfg = M.MVLBertPretrainConfig(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=1024,
                                  vocab_size=V, ITM_task=True)
    cfg.swin.update(embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], drop_path_rate=0.2)
    cfg.ITM_task = True
    model = M.MVLBertForPretraining(cfg).cuda().train()
    step = PretrainStep(model, lr=1e-4)
    seen = []
    for world_rank in (0, 1):
        it = PretrainBatches(Shard(str(tmp_path / "s0")), 3, "cuda", num_replicas=2, rank=world_rank, seed=5, vocab_size=V)
        n = 0
        for image, masked, labels, itm, lengths in it:
            assert image.shape == (3, 3, 224, 224) and image.dtype == torch.float32 and image.is_cuda
            assert masked.shape == (3, T) and labels.shape == (3, T) and itm.shape == (3,)
            assert not lengths.is_cuda and lengths.dtype == torch.int32
            assert torch.equal(lengths, (masked != 0).sum(1).cpu().to(torch.int32))
            loss = step((image, masked, labels, itm, lengths))
            # a batch whose samples are all ITM negatives has no MLM label: mean over nothing = NaN, in the
            # reference (F.cross_entropy, ignore_index) as well
            assert torch.isfinite(loss) or bool((labels < 0).all())
            n += 1
        seen.append(n)
    assert seen == [2, 2]            # 12 samples / 2 ranks / batch 3
