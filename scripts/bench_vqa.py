"""BASELINE config #1 shape on the GPU: SLAKE Med-VQA forward, Swin-S + BERT-base, B=2, T=80 (and T=23)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvlt_amd as M
torch.manual_seed(0)
for cd, graph in ((torch.bfloat16, False), (torch.bfloat16, True), (torch.float32, False), (torch.float32, True)):
    cfg = M.MVLBertConfigforVQA()
    cfg.eval_cuda_graph = graph
    model = M.set_compute_dtype(M.MVLBertForVQA(cfg).cuda().eval(), cd)
    for B, T in ((2, 80), (2, 23), (32, 80)):
        img = torch.randn(B, 3, 224, 224, device="cuda"); q = torch.randint(1000, 30000, (B, T), device="cuda")
        with torch.no_grad():
            for _ in range(5): model(img, q, None)
            torch.cuda.synchronize(); t = time.time()
            n = 30
            for _ in range(n): prob, logits = model(img, q, None)
            torch.cuda.synchronize(); dt = (time.time() - t) / n
        print(f"VQA forward {'replayed graph' if graph else 'eager':14s} {str(cd).split('.')[-1]:8s} B={B:2d} T={T:2d}: {dt*1e3:6.2f} ms  ({B/dt:7.1f} samples/s)", flush=True)
