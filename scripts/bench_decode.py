"""Config #4 shape: greedy report generation, Swin-S + BERT-base, B=32, max_length=150 (MIMIC-CXR), bf16."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvlt_amd as M
torch.manual_seed(0)
cfg = M.MVLBertConfigForImageCaption(); cfg.max_length = int(os.environ.get("MAXLEN", 150)); cfg.eos_token_id = None   # never stop early: fixed work
tok = type("Tok", (), {"mask_token_id": 103, "sep_token_id": 102})()
model = M.MVLBertForImageCaption(cfg, tokenizer=tok).cuda().eval()
img = torch.randn(32, 3, 224, 224, device="cuda")
for _ in range(2):
    ids, _ = model(img, None, 1, "unilm")
torch.cuda.synchronize(); t = time.time()
n = 3
for _ in range(n):
    ids, _ = model(img, None, 1, "unilm")
torch.cuda.synchronize(); dt = (time.time() - t) / n
print(f"decode B=32 max_length={cfg.max_length}: {dt*1e3:.1f} ms/batch, {32/dt:.1f} reports/s, {32*ids.shape[1]/dt:.0f} tokens/s, {dt/ids.shape[1]*1e3:.2f} ms/step", flush=True)
