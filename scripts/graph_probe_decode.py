"""Feasibility probe: one cached 2-token decode step captured in a HIP graph vs eager (timing only;
`past` is baked into the captured launches)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvlt_amd as M
from mvlt_amd import decode as D, ops
from mvlt_amd.arena import Arena
torch.manual_seed(0)
cfg = M.MVLBertConfigForImageCaption(); cfg.max_length = 150; cfg.eos_token_id = None
tok = type("Tok", (), {"mask_token_id": 103, "sep_token_id": 102})()
model = M.MVLBertForImageCaption(cfg, tokenizer=tok).cuda().eval()
mv = model.MVLBert
cd = torch.bfloat16
ar = Arena.of(model, cd); ar.refresh_shadow()
B, H, nH, hd, nl = 32, 768, 12, 64, 12
cap = 51 + 151
kc = [torch.zeros((B, nH, cap, hd), dtype=cd, device="cuda") for _ in range(nl)]
vc = [torch.zeros((B, nH, cap, hd), dtype=cd, device="cuda") for _ in range(nl)]
new_ids = torch.randint(1000, 30000, (B, 2), device="cuda")
head = model.MLM_head_seq2seq
V = head.predictions.decoder.out_features
past = 100

def step():
    x = D._embed_new(mv, new_ids, past, cd).view(B * 2, H)
    h = D._layers_cached(mv, ar, x, kc, vc, past, 2).view(B, 2, H)
    pre, t1, t2, _, _ = head._transform(ar, h[:, -1].contiguous(), False)
    logits, _ = head._logits(ar, t2)
    return ops.argmax(logits, V)

with torch.no_grad():
    for _ in range(3): step()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(50): step()
    torch.cuda.synchronize(); e = (time.time() - t) / 50
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): step()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(50): g.replay()
    torch.cuda.synchronize(); r = (time.time() - t) / 50
print(f"decode step: eager {e*1e3:.3f} ms, graph replay {r*1e3:.3f} ms", flush=True)
