"""RCCL all-reduce bus bandwidth probe at the gradient-bucket sizes (25 / 64 / 400 MB, f32 and bf16).
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/rccl_busbw.py
busbw = 2 (N-1)/N x bytes / time (ring model; xGMI is point-to-point, 7 links x ~153 GB/s per GPU)."""
import os
import torch
import torch.distributed as dist
rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
for dt in (torch.float32, torch.bfloat16):
    for mb in (25, 64, 400):
        n = mb * (1 << 20) // (4 if dt == torch.float32 else 2)
        x = torch.ones(n, dtype=dt, device="cuda")
        for _ in range(3):
            dist.all_reduce(x)
        torch.cuda.synchronize(); dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dist.all_reduce(x)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        if rank == 0:
            by = n * x.element_size()
            print(f"{str(dt):15s} {mb:4d} MB: {ms*1e3:8.1f} us  algbw {by/ms/1e6:7.1f} GB/s  busbw {2*(world-1)/max(world,1)*by/ms/1e6:7.1f} GB/s  (world {world})", flush=True)
dist.destroy_process_group()
