"""Does running two half-batch dgrad chains on two streams beat one full-batch chain?  Proxy: the main-stream kernels of
18 Swin stage-2 block backward passes (fc2 dgrad + GELU', fc1 dgrad, LayerNorm bwd, proj dgrad, window-attention bwd, qkv
dgrad, LayerNorm bwd; no weight gradients), captured in HIP graphs so the host is out of the picture.
    (a) one graph, B = 32                    (b) two graphs, B = 16 each, replayed on two streams at once
python scripts/dual_chain_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvlt_amd  # noqa
from mvlt_amd import ops, _lib as L
from mvlt_amd.indexing import batched_window_maps

dev = torch.device("cuda:0")
C, nH, H, ws, NB = 384, 12, 14, 7, 18
dt = torch.bfloat16


def make(B):
    R = B * H * H
    r = lambda *s: (torch.randn(*s, device=dev) * 0.1).to(dt)
    st = dict(B=B, R=R, dx=r(R, C), h=r(R, 4 * C), x1=r(R, C), x=r(R, C), qkv=r(R, 3 * C), ao=r(R, C),
              w1=r(4 * C, C), w2=r(C, 4 * C), wp=r(C, C), wq=r(3 * C, C), g=torch.ones(C, device=dev),
              dg=torch.zeros(C, device=dev), db=torch.zeros(C, device=dev), table=torch.zeros(169, nH, device=dev),
              dtab=torch.zeros(169, nH, device=dev))
    st["mean"] = torch.zeros(R, device=dev); st["rstd"] = torch.ones(R, device=dev)
    nW = (H // ws) ** 2
    st["lse"] = torch.zeros(B * nW, nH, ws * ws, device=dev) + 3.9
    st["maps"] = batched_window_maps(B, H, H, ws, 3, dev)
    st["q"] = ops.LnReduceQueue()
    return st


def block_bwd(st, dx2):
    w2n, n2w = st["maps"]
    B = st["B"]
    nW = (H // ws) ** 2
    q = st["q"]
    dh = ops.gemm(dx2, st["w2"], b_kmajor=True, mul_gelu_grad=st["h"])
    dxn2 = ops.gemm(dh, st["w1"], b_kmajor=True)
    dx1, dyw = ops.layernorm_bwd(dxn2, st["x1"], st["mean"], st["rstd"], st["g"], st["dg"], st["db"], dres=dx2,
                                 branch=dict(rowmap=n2w), defer=q)
    dao = ops.gemm(dyw, st["wp"], b_kmajor=True)
    dqkv = ops.attn_bwd(dao, st["qkv"], st["ao"], st["lse"], L.ATTN_SWIN, B * nW, ws * ws, nH, C // nH, (C // nH) ** -0.5,
                        dbias_table=st["dtab"], bias_table=st["table"], nW=nW, win_res=H, shift=3)
    dxn1w = ops.gemm(dqkv, st["wq"], b_kmajor=True)
    dx0 = ops.layernorm_bwd(dxn1w, st["x"], st["mean"], st["rstd"], st["g"], st["dg"], st["db"], dy_rowmap=n2w, dres=dx1, defer=q)
    q.items.clear(); q.off = 0
    return dx0


def chain(st):
    dx = st["dx"]
    for _ in range(NB):
        dx = block_bwd(st, dx)
    return dx


def capture(st, stream):
    with torch.cuda.stream(stream):
        with ops.pin_stream():
            chain(st)                                   # warm-up: workspaces, lazy attributes
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        with ops.pin_stream():
            st["out"] = chain(st)
    return g


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
full = make(32); ga = capture(full, s1)
ha, hb = make(16), make(16)
gb1 = capture(ha, s1); gb2 = capture(hb, s2)
cur = torch.cuda.current_stream()


def run_full():
    s1.wait_stream(cur)
    with torch.cuda.stream(s1): ga.replay()
    cur.wait_stream(s1)


def run_dual():
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): gb1.replay()
    with torch.cuda.stream(s2): gb2.replay()
    cur.wait_stream(s1); cur.wait_stream(s2)


def run_half_serial():
    s1.wait_stream(cur)
    with torch.cuda.stream(s1): gb1.replay(); 
    cur.wait_stream(s1)


ta, tb, tc = timeit(run_full), timeit(run_dual), timeit(run_half_serial)
print(f"(a) one chain, B=32: {ta * 1e3:8.1f} us for {NB} blocks = {ta * 1e3 / NB:6.1f} us per block")
print(f"(b) two chains, B=16 + B=16 on two streams: {tb * 1e3:8.1f} us = {tb * 1e3 / NB:6.1f} us per block pair")
print(f"(c) one chain, B=16 alone: {tc * 1e3:8.1f} us = {tc * 1e3 / NB:6.1f} us per block")
