"""Stand-alone timing of the grouped weight-gradient launch (mvlt_gemm_group) on the step's layer shapes (B=32):
    python scripts/bench_wgrad_group.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvlt_amd  # noqa
from mvlt_amd import ops
dev = torch.device("cuda:0")
LAYERS = {"bert layer (R=3150)": (3150, [(768, 3072), (3072, 768), (768, 2304), (768, 768)]),
          "bert layer (R=4192)": (4192, [(768, 3072), (3072, 768), (768, 2304), (768, 768)]),
          "swin s2 block (R=6272)": (6272, [(384, 1536), (1536, 384), (384, 1152), (384, 384)]),
          "swin s3 block (R=1568)": (1568, [(768, 3072), (3072, 768), (768, 2304), (768, 768)]),
          "swin s1 block (R=25088)": (25088, [(192, 768), (768, 192), (192, 576), (192, 192)]),
          "swin s0 block (R=100352)": (100352, [(96, 384), (384, 96), (96, 288), (96, 96)])}
ONLY = os.environ.get("ONLY")          # e.g. ONLY="s0 block,s1 block"
for name, (R, shp) in LAYERS.items():
    if ONLY and not any(o in name for o in ONLY.split(",")): continue
    items, fl = [], 0.0
    for ni, no in shp:
        dy = torch.randn(R, no, device=dev).bfloat16(); x = torch.randn(R, ni, device=dev).bfloat16()
        items.append((dy, x, torch.empty(no, ni, device=dev), None if os.environ.get("NO_DB") else torch.empty(no, device=dev)))
        fl += 2.0 * R * ni * no
    for _ in range(3): ops.wgrad_group(items)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): ops.wgrad_group(items)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"{name:28s} {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s", flush=True)
