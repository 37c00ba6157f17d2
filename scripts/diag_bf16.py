"""Diagnostic (GPU): bf16 path vs exact-f32 path of the SAME HIP kernels, on formula
and on randomly initialised weights; prints global relative gradient errors."""
import copy, json, os, random, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import mvlt_amd as M
from conftest import formula_sd, synth_batch

def run(model, cd, batch, flip):
    M.set_compute_dtype(model, cd)
    for p in model.parameters(): p.grad = None
    random.seed(0)
    orig = random.random
    random.random = lambda: flip
    try:
        loss = model(*batch)
    finally:
        random.random = orig
    loss.backward(); torch.cuda.synchronize()
    return loss.item(), {k: p.grad.double().clone() for k, p in model.named_parameters() if p.grad is not None}

def compare(tag, model, batch):
    for flip in (0.1, 0.9):
        l32, g32 = run(model, torch.float32, batch, flip)
        l16, g16 = run(model, torch.bfloat16, batch, flip)
        num = sum(float((g16[k] - g32[k]).pow(2).sum()) for k in g32)
        den = sum(float(g32[k].pow(2).sum()) for k in g32)
        worst = sorted(((float((g16[k]-g32[k]).norm()/(g32[k].norm()+1e-30)), k) for k in g32 if g32[k].norm() > 1e-3*den**0.5), reverse=True)[:4]
        print(f"{tag} flip={flip}: loss f32={l32:.6f} bf16={l16:.6f} rel={(l16-l32)/l32:+.2e}  grad global rel={ (num/den)**0.5:.4f}  worst={[(round(a,3),k) for a,k in worst]}", flush=True)

specs = json.load(open("tests/golden/specs.json"))
torch.manual_seed(0)
# tiny
cfg = M.MVLBertPretrainConfig(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=1024, vocab_size=3000)
cfg.swin.update(embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], drop_path_rate=0.2); cfg.ITM_task = True
image, ids, labels, itm = synth_batch(3, 24, seed=41, vocab=3000)
batch = (image.cuda(), ids.cuda(), labels.cuda(), itm.cuda())
m = M.MVLBertForPretraining(cfg).cuda().eval()
compare("tiny random-init", m, batch)
m.load_state_dict(formula_sd(specs["tiny_pretrain"]), strict=False)
compare("tiny formula    ", m, batch)
# full
cfg = M.MVLBertPretrainConfig(); cfg.ITM_task = True
image, ids, labels, itm = synth_batch(2, 80, seed=21)
batch = (image.cuda(), ids.cuda(), labels.cuda(), itm.cuda())
m = M.MVLBertForPretraining(cfg).cuda().eval()
compare("full random-init", m, batch)
m.load_state_dict(formula_sd(specs["pretrain"]), strict=False)
compare("full formula    ", m, batch)
