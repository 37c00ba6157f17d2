"""Worker of tests/test_ddp_gpu.py: one data-parallel rank of the real (tiny) pre-training model on the GPU.
    RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment; argv: backend device_index out_file average overlap
Each rank takes its slice of a fixed batch, runs forward + backward with mvlt_amd.ddp.GradReducer (average = 1: rank
mean, 0: sum; overlap = 1: FusedAdamW consumes the buckets beside the backward pass) and rank 0 stores the reduced
gradients, its loss and -- with overlap -- the updated parameters."""
import os
import random
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build_model(M):
    from conftest import hash_sd
    import json
    with open(os.path.join(ROOT, "tests", "golden", "specs_hash.json")) as f:
        spec = json.load(f)["hash_tiny_pretrain"]
    cfg = M.MVLBertPretrainConfig(hidden_size=256, num_hidden_layers=2, num_attention_heads=4,
                                  intermediate_size=1024, vocab_size=3000)
    cfg.swin.update(embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], drop_path_rate=0.2)
    cfg.ITM_task = True
    model = M.MVLBertForPretraining(cfg)
    model.load_state_dict(hash_sd(spec), strict=False)
    return M.set_compute_dtype(model.cuda().eval(), torch.float32)


def main():
    backend, dev, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    average = len(sys.argv) > 4 and sys.argv[4] == "1"
    overlap = len(sys.argv) > 5 and sys.argv[5] == "1"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    import mvlt_amd as M
    from mvlt_amd.ddp import GradReducer
    from conftest import synth_batch
    model = build_model(M)
    red = GradReducer(model, bucket_bytes=256 << 10, average=average)
    image, ids, labels, itm = synth_batch(2 * world, 24, seed=71, vocab=3000)
    sl = slice(2 * rank, 2 * rank + 2)
    random.random = lambda: 0.9
    batch = (image[sl].cuda(), ids[sl].cuda(), labels[sl].cuda(), itm[sl].cuda())
    params_after = None
    if overlap:
        from mvlt_amd.train import PretrainStep
        step = PretrainStep(model, lr=1e-3, reducer=red, world_size=world, overlap_optimizer=True)
        loss = step(batch)
        torch.cuda.synchronize()
        params_after = {k: p.detach().cpu().clone() for k, p in model.named_parameters()}
    else:
        loss = model(*batch)
        loss.backward()
    torch.cuda.synchronize()
    assert len(red.launched) > 2
    if rank == 0:
        torch.save({"grads": {k: p.grad.detach().cpu() for k, p in model.named_parameters() if p.grad is not None},
                    "loss": float(loss), "params_after": params_after}, out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
