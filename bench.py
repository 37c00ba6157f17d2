"""Headline benchmark: image-text pairs/sec of one MVLT pre-training step
(Swin-S + BERT-base, 224 px, seq 80, MLM+ITM, bf16) on N MI355X GPUs.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = forward + backward (+ RCCL gradient all-reduce) + AdamW over one
synthetic batch that is already resident in HBM (B=32 per GPU: weak scaling).
Rank 0 prints ONE JSON line with `roofline` (dominant kernel, timed live with
HIP events on the launch stream) and `cpu_baseline` (the oracle timed on the
host cores, N=1 only).
"""
import argparse
import json
import os
import random
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # before the HIP runtime starts (mvlt_amd/__init__.py says why)

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_PAIR = 132.7          # fwd 44.22 GFLOP x 3 (reference-equivalent work)
PEAK_BF16_TFLOPS = 2500.0       # MI355X_MICROARCH.md: dense bf16 MFMA peak
PER_GPU_BATCH, SEQ = 32, 80


DOMINANT_NAME = ("forward + dgrad GEMM family of the Swin blocks / BertLayers: gemm_kernel<bf16,{128|64},{128|96|64},row,{row|kmajor}>, "
                 "gemm_glds_kernel<{64|128},{64|96|128}>, gemm8_kernel, rowstream_kernel (x W^T and dy W with fused epilogues; main "
                 "stream) -- the family with the largest share of GPU time (~45 %, profiles/r6_bench_kernel_stats.csv)")
WGRAD_NAME = ("gemm_group_glds_kernel<{128|64},128,{2|3}> (LDS-DMA, both operands k-major; round 6) and gemm_group_kernel<bf16,64,96,kmajor,kmajor> / "
              "gemm8_kernel (Swin stages 0 / 1): grouped weight-gradient GEMMs dW_i = dY_i^T X_i + bias gradients of one layer per "
              "launch; side stream, beside the dgrad chain")
DOMINANT = ("group", 1, 64, 128)   # gemm_group_kernel<bf16, BM=128|64, BN=128, A k-major, B k-major>: the grouped weight-
                                   # gradient GEMM (all dW of one BertLayer / Swin block per launch), the symbol with
                                   # the largest share of GPU time (profiles/r1_bench_kernel_stats.csv)


class KernelTimer:
    """Brackets launches of ONE kernel symbol with HIP events on the stream it is launched on.
    Every ``every``-th launch is sampled: an event pair per launch costs host time and a barrier
    packet on the stream (measured: 19.3 vs 18.1 ms/step with all ~140 launches/step bracketed)."""

    def __init__(self, key, every=8):
        self.key, self.pairs, self.flops = key, [], 0.0
        self.layout = (key[3], key[4]) if key[0] != "group" else None      # single-GEMM keys: operand layout watched
        self.every, self.seen = every, 0

    def __call__(self, flops, key):
        if key != self.key:
            return None
        self.seen += 1
        if self.seen % self.every:
            return None
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.pairs.append((s, e))
        self.flops += flops
        return s, e

    def result(self):
        if not self.pairs:
            return None
        ms = sum(s.elapsed_time(e) for s, e in self.pairs)
        return dict(launches=len(self.pairs), avg_us=1e3 * ms / len(self.pairs), tflops=self.flops / (ms * 1e-3) / 1e12)


def _cpu_threads():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 64))


def cpu_baseline_worker():
    """Child process: the CPU oracle (plain fp32 PyTorch restatement of the reference, train mode
    with the reference dropouts) timed on this host's cores: fwd + bwd + AdamW, B=8."""
    from oracle import mvlt_oracle as O
    from mvlt_amd.train import synthetic_batch
    import mvlt_amd as M
    threads = _cpu_threads()
    torch.set_num_threads(threads)
    B = int(os.environ.get("MVLT_CPU_BASELINE_B", "8"))
    model = M.MVLBertForPretraining(M.MVLBertPretrainConfig())        # parameter container only (CPU); math = oracle
    sd = {k: (v.detach().clone().requires_grad_(True) if v.dtype.is_floating_point else v)
          for k, v in model.state_dict().items()}
    del model
    params = [v for v in sd.values() if v.dtype.is_floating_point]
    opt = torch.optim.AdamW(params, lr=4e-5, betas=(0.9, 0.999), eps=1e-6, weight_decay=1e-4)
    image, ids, labels, itm = synthetic_batch(B, SEQ, "cpu", 99)
    drop = O.Dropper("torch")
    times = []
    t_start = time.time()
    for i in range(4):
        t = time.time()
        loss = O.pretrain_loss(sd, O.SwinCfg(), O.BertCfg(), image, ids, labels, itm, i % 2 == 0, itm_task=True, drop=drop)
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        times.append(time.time() - t)
        if time.time() - t_start > 45.0:
            break
    timed = times[1:] if len(times) > 1 else times          # first step = warm-up when there is more than one
    print(json.dumps(dict(value=round(B * len(timed) / sum(timed), 3), unit="pairs/s", cores=threads, kind="port",
                          sample=f"{len(timed)} timed step(s) of B={B}, T={SEQ}, fwd+bwd+AdamW, fp32 train mode "
                                 f"(oracle/mvlt_oracle.py, {len(times) - len(timed)} warm-up)")), flush=True)


def cpu_baseline(timeout_s=170):
    """Runs the worker as a CHILD process with a hard timeout so the benchmark always finishes."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], capture_output=True,
                           text=True, timeout=timeout_s, env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return dict(value=None, unit="pairs/s", cores=_cpu_threads(), kind="port", sample="worker produced no result: " + r.stderr[-200:])
    except subprocess.TimeoutExpired:
        return dict(value=None, unit="pairs/s", cores=_cpu_threads(), kind="port", sample=f"timed out after {timeout_s}s")


def self_launch(args):
    """`python bench.py --gpus N` outside a launcher: start the N ranks as CHILD processes (torch.distributed.run)
    before this process has touched the GPU, relay rank 0's JSON line and exit with the children's code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run(cmd, env=env, capture_output=True, text=True)
    line = next((ln for ln in reversed(r.stdout.splitlines()) if ln.startswith('{"metric"')), None)
    if line is None:
        sys.stderr.write(r.stdout[-4000:] + "\n" + r.stderr[-4000:] + "\n")
        raise SystemExit(r.returncode or 1)
    print(line, flush=True)
    raise SystemExit(r.returncode)


def packed_gflop_per_pair(batch, all_mlm_rows=False):
    """FLOPs actually executed per pair by the packed-rows / labelled-rows variant (SURVEY 8d formulas with the
    BERT rows sum(51 + len_b) instead of B*131 and the MLM head on 10 rows per sample instead of 80)."""
    lens = batch[4].float() + 51.0
    B = lens.numel()
    L = 51 + SEQ
    bert_dense = 12 * (L * (4 * 768 ** 2 + 2 * 768 * 3072) + 2 * L * L * 768)
    bert_packed = float(sum(12 * (l * (4 * 768 ** 2 + 2 * 768 * 3072) + 2 * l * l * 768) for l in lens.tolist())) / B
    head_all = SEQ * (768 ** 2 + 768 * 30522)
    head_lab = (SEQ if all_mlm_rows else 10) * (768 ** 2 + 768 * 30522)
    fwd_ref = GFLOP_PER_PAIR / 3 / 2 * 1e9           # MACs of the reference-equivalent forward
    fwd = fwd_ref - bert_dense + bert_packed - head_all + head_lab
    return fwd * 2 * 3 / 1e9


def timed_run(step, batch, steps, use_dist, dist, blocks=5):
    """EXACTLY `steps` steps between two barrier + synchronize brackets (max over ranks).  Beside the total, the stream
    is stamped with an event every steps/blocks steps (recorded, never waited for inside the region), so the per-block
    times of the same run can be quoted: median / min / max of `blocks` blocks."""
    per = max(1, steps // blocks) if steps >= 2 * blocks else 0
    stamps = []
    # (a deferred optimizer tail of the last warm-up step is applied BEFORE the region starts and the last timed step's tail
    # INSIDE it: the region holds the complete work of exactly `steps` steps -- PretrainStep(defer_optimizer_tail=True))
    flush = getattr(step, "flush", None)
    if flush is not None:
        flush()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if per:
        e = torch.cuda.Event(enable_timing=True); e.record(); stamps.append((0, e))
    for i in range(steps):
        loss = step(batch)
        if per and (i + 1) % per == 0 and (i + 1) // per <= blocks:
            e = torch.cuda.Event(enable_timing=True); e.record(); stamps.append((i + 1, e))
    if flush is not None:
        flush()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    timed_run.block_ms = [a[1].elapsed_time(b[1]) / (b[0] - a[0]) for a, b in zip(stamps, stamps[1:])]
    return elapsed, loss


def block_stats(per_gpu_batch, world):
    """{"blocks": n, "ms_per_step_median/min/max", "value_median/min/max"} of the last timed_run (this rank's stream)."""
    b = sorted(getattr(timed_run, "block_ms", []))
    if not b:
        return None
    med = b[len(b) // 2] if len(b) % 2 else 0.5 * (b[len(b) // 2 - 1] + b[len(b) // 2])
    pps = lambda ms: round(per_gpu_batch * world / (ms * 1e-3), 1)
    return {"blocks": len(b), "ms_per_step_median": round(med, 3), "ms_per_step_min": round(b[0], 3), "ms_per_step_max": round(b[-1], 3),
            "value_median": pps(med), "value_min": pps(b[-1]), "value_max": pps(b[0]),
            "note": "device time between events recorded on the compute stream every steps/5 steps inside the timed region "
                    "(rank 0; `value` itself is steps / wall time of the whole region, max over ranks)"}


def other_configs(M):
    """BASELINE configs #4 and #5 at their stated sizes, one GPU, reported as extra keys of the same line (the
    headline stays config #2).  #4: greedy report generation, Swin-S + BERT-base, B=32, max_length=150 (MIMIC-CXR
    shape, run_report_generation_cxr.py:385-386), bf16, replayed HIP graph.  #5: Swin-B + BERT-base pretrain step,
    per-GPU batch 8 (global 64 on 8 GPUs), seq 128."""
    from mvlt_amd.train import PretrainStep, synthetic_batch
    res = {}
    torch.manual_seed(0)
    cfg = M.MVLBertConfigForImageCaption()
    cfg.max_length, cfg.eos_token_id = 150, None                      # fixed work: never stop early
    tok = type("Tok", (), {"mask_token_id": 103, "sep_token_id": 102})()
    cap = M.MVLBertForImageCaption(cfg, tokenizer=tok).cuda().eval()
    img = torch.randn(32, 3, 224, 224, device="cuda")
    for _ in range(2):
        cap(img, None, 1, "unilm")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        ids, _ = cap(img, None, 1, "unilm")
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    res["config4_decode"] = {"workload": "greedy decode, Swin-S + BERT-base, B=32, max_length=150, bf16, KV cache, HIP graph replay",
                             "ms_per_batch": round(dt * 1e3, 2), "reports_per_s": round(32 / dt, 1),
                             "tokens_per_s": round(32 * ids.shape[1] / dt, 0),
                             "us_per_2token_step": round(dt / ids.shape[1] * 1e6, 1),
                             # per 2-token step: 217 MB of bf16 weights + the valid K / V rows of the cache (12 layers x 2 x
                             # B x (51 + t) rows x 768 x 2 B, t averaged over the 150 steps: ~149 MB) at the ~5.5 TB/s this chip streams
                             # reads at (profiles/r5_hbm_write_probe.txt); the weights alone at the nominal 8 TB/s would be 27 us
                             "hbm_floor_us_per_step": round((217e6 + 12 * 2 * 32 * (51 + (ids.shape[1] - 1) / 2.0) * 768 * 2) / 5.5e12 * 1e6, 1),
                             "note": "floor = 217 MB of bf16 weights + ~149 MB of valid K/V cache rows per step at the measured ~5.5 TB/s read "
                                     "rate (weights alone at the nominal 8 TB/s: 27 us); the step is a chain of ~100 dependent small "
                                     "kernels, i.e. latency bound"}
    del cap
    # config #1: SLAKE Med-VQA forward, B = 2, T = 80 / 23 (the reference's CPU-runnable case, run_vqa.py): latency of one call
    vq = M.MVLBertForVQA(M.MVLBertConfigforVQA()).cuda().eval()
    c1 = {}
    with torch.no_grad():
        for T in (80, 23):
            im, q = torch.randn(2, 3, 224, 224, device="cuda"), torch.randint(1000, 30000, (2, T), device="cuda")
            for _ in range(5):
                vq(im, q, None)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30):
                vq(im, q, None)
            torch.cuda.synchronize()
            c1[f"ms_per_call_T{T}"] = round((time.perf_counter() - t0) / 30 * 1e3, 3)
    c1["workload"] = ("Med-VQA forward (MVLBertForVQA.forward), Swin-S + BERT-base, batch 2, 224 px, question length 80 / 23, bf16, eval: "
                      "~330 dependent launches, device-latency bound (a replayed HIP graph of the call, config.eval_cuda_graph, takes the same time)")
    res["config1_vqa_forward"] = c1
    del vq
    c5 = M.MVLBertPretrainConfig().use_swin_base()
    c5.ITM_task = True
    m5 = M.MVLBertForPretraining(c5).cuda().train()
    st5 = PretrainStep(m5)
    b5 = synthetic_batch(8, 128, "cuda", 77)[:4]
    for _ in range(4):
        st5(b5)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        st5(b5)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    res["config5_swin_b"] = {"workload": "pretrain step (MLM+ITM), Swin-B [2,2,18,2] + added Linear(1024,768) + BERT-base, batch 8/GPU, seq 128, bf16",
                             "ms_per_step": round(dt * 1e3, 3), "pairs_per_s": round(8 / dt, 1),
                             "step_tflops_reference_equivalent": round(8 / dt * 206.1 / 1e3, 1)}
    del m5, st5
    torch.cuda.empty_cache()
    return res


def one_rank_rccl(steps=40, warmup=10):
    """the same step with the gradient reducer and RCCL on ONE rank (MVLT_FORCE_DDP=1), in a child process.
    What a 1-GPU lease can show of the multi-GPU path: bucket issue, stream joins and RCCL's kernel beside the backward pass.
    MVLT_DDP_NULL_COLLECTIVE=1 is the same run without the collective itself: the difference to the plain step is the reducer's
    own cost, the rest is RCCL's one-rank pass over the 836 MB gradient arena (a copy a real ring does not make)."""
    import subprocess
    out = {}
    arms = (("ms_per_step", {}), ("ms_per_step_without_the_collective", {"MVLT_DDP_NULL_COLLECTIVE": "1"}),
            # round 6: the same run with the end-of-backward wait deferred (what is left is the collective's kernel BESIDE the
            # backward pass: the exposed tail is the difference to the first figure), and a rehearsal of world size 8 -- every
            # bucket copied twice by 32 persistent workgroups at a few hundred GB/s on a third stream, the shape of a ring
            # all-reduce that is xGMI-bound for milliseconds (mvlt_amd/ddp.py: MVLT_DDP_DEFER_WAIT, MVLT_DDP_REHEARSE)
            ("ms_per_step_wait_deferred", {"MVLT_DDP_DEFER_WAIT": "1"}),
            ("ms_per_step_ring_rehearsal_64wg", {"MVLT_DDP_REHEARSE": "64,4"}))
    for key, extra in arms:
        env = dict(os.environ, MVLT_FORCE_DDP="1", **extra)
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--steps", str(steps), "--warmup", str(warmup),
                                "--no-cpu-baseline", "--no-extra"], env=env, capture_output=True, text=True, timeout=240)
            line = next((ln for ln in reversed(r.stdout.splitlines()) if ln.startswith('{"metric"')), None)
            out[key] = json.loads(line)["ms_per_step"] if line else None
        except Exception as e:          # the headline must not depend on this extra
            out[key] = None
            out["error"] = repr(e)[:200]
    out["workload"] = ("config #2 step with GradReducer over RCCL on one rank (MVLT_FORCE_DDP=1): 64 MiB buckets (8 MiB for the last 32 MB "
                       "of the arena) exchanged one bucket late on the main stream; second figure: reducer without the collective; third: "
                       "collective issued, end-of-backward wait deferred; fourth: ring-collective rehearsal (2 x bucket bytes, 64 workgroups, ~400 GB/s)")
    return out


def profiled_traffic(key):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 PMC passes (cannot be collected inside this
    process): profiles/r6_dominant_kernel_traffic.json {"family": {...}, "wgrad_group": {...}}, filled in from the
    scripts/pmc.py passes over scripts/profile_step.py (2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md HBM section)."""
    name = "profiles/r6_dominant_kernel_traffic.json"
    try:
        with open(os.path.join(ROOT, name)) as fh:
            d = json.load(fh)
        return d[key].get("traffic_bytes_per_launch"), name
    except Exception:
        return None, None


def roofline_entry(samples, name, traffic_key, every):
    """samples: [(executed flops, ms, algorithmic bytes)] of launches bracketed with HIP events inside the timed region."""
    if not samples:
        return None
    ms = sum(s[1] for s in samples)
    tflops = sum(s[0] for s in samples) / (ms * 1e-3) / 1e12
    traffic, src = profiled_traffic(traffic_key)
    algo = sum(s[2] for s in samples) / len(samples) if len(samples[0]) > 2 else None
    return {"bound": "mfma", "kernel": name, "achieved": round(tflops, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tflops / PEAK_BF16_TFLOPS, 4),
            # HBM-side bytes per launch: rocprofv3 PMC passes of the same step (2*FETCH_SIZE + WRITE_SIZE), read from the
            # committed profile file named in traffic_source -- not measurable in-process
            "traffic": traffic, "traffic_source": src,
            # algorithmic bytes per launch of the SAME sampled launches: weights once, activation rows in, output rows out,
            # epilogue operands (residual / saved pre-activation); with the row count the kernels read on the device
            "algorithmic_bytes": None if algo is None else int(algo),
            "traffic_over_algorithmic": None if (algo is None or not traffic) else round(traffic / algo, 2),
            "launches": len(samples), "avg_launch_us": round(1e3 * ms / len(samples), 2),
            "flops": "executed: 2 M N K with the row count the kernels read on the device (ragged batches), not the dense bound",
            "sampling": f"1 in {every} launches of the family inside the timed region, HIP events on the launch stream"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)      # >= 100 timed steps
    ap.add_argument("--warmup", type=int, default=20)     # ... after >= 20 warm-up steps
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the packed-rows / labelled-rows extra measurement")
    args = ap.parse_args()
    if args.cpu_baseline_only:
        cpu_baseline_worker()
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        self_launch(args)                           # never returns; this process has not touched the GPU
    import torch.distributed as dist
    # MVLT_BENCH_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (the ranks then share the
    # devices round-robin and exchange through gloo; RCCL refuses two ranks on one device).  tests/test_bench_gpu.py runs
    # the self-launch -> torchrun -> relay path this way; the driver's SCALE runs use the default (RCCL, one rank per GPU).
    backend = os.environ.get("MVLT_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > ndev:
        raise SystemExit(f"bench.py: {world} ranks over RCCL need {world} GPUs, this box has {ndev} "
                         "(MVLT_BENCH_BACKEND=gloo shares the devices for a functional rehearsal)")
    local = local % max(1, ndev) if backend != "nccl" else local
    torch.cuda.set_device(local)
    use_dist = world > 1 or os.environ.get("MVLT_FORCE_DDP") == "1"     # FORCE: exercise RCCL + reducer on 1 rank
    real_stdout = None
    if use_dist:
        # RCCL prints a version banner on stdout when the communicator is created; stdout must carry ONE JSON line,
        # so file descriptor 1 points at stderr for the run and the line goes out through a saved duplicate
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import mvlt_amd as M
    from mvlt_amd import ops
    from mvlt_amd.ddp import GradReducer, seed_coin_flip
    from mvlt_amd.train import PretrainStep, synthetic_batch

    torch.manual_seed(1234)                         # same random-init replica on every rank
    cfg = M.MVLBertPretrainConfig()
    cfg.ITM_task = True                             # BASELINE config: Pretrain (MLM+ITM)
    # PRIMARY number = the reference's own call: model(image, caption_masked, caption_label, ITM_label)
    # (modules/model.py:372) with the module defaults: the packing plan for the zero-padded caption tails and the
    # labelled-rows-first MLM head are derived on the device from the ids / labels themselves (no new argument, no host
    # sync; same loss and gradients as computing every row, which is reported beside it as value_dense_rows).
    cfg.mlm_max_labels_per_sample = None
    model = M.MVLBertForPretraining(cfg).cuda().train()
    M.manual_seed(4321 + rank)                      # dropout stream differs per rank
    seed_coin_flip(5678)                            # seq2seq/bidir flip identical on all ranks
    # MVLT_DDP_BF16=1: bf16-compressed gradient exchange (non-default; the reference-equivalent sum is f32)
    comm = torch.bfloat16 if os.environ.get("MVLT_DDP_BF16") == "1" else torch.float32
    bucket_mb = int(os.environ.get("MVLT_DDP_BUCKET_MB", "64"))
    reducer = GradReducer(model, comm_dtype=comm, bucket_bytes=bucket_mb << 20) if use_dist else None
    # MVLT_DEFER_OPT_TAIL=1 (A/B switch, off by default): the AdamW sweep over BertLayers 1.. and the heads runs beside the NEXT
    # step's encoder forward (mvlt_amd/optim.py).  Measured 11.41 / 11.41 / 11.43 against 11.46 / 11.47 / 11.40 ms: inside the box's
    # noise -- the encoder forward gives back what the sweep gains (profiles/HISTORY.md, round 6)
    defer_tail = os.environ.get("MVLT_DEFER_OPT_TAIL", "0") == "1"
    step = PretrainStep(model, reducer=reducer, world_size=world, defer_optimizer_tail=defer_tail)
    lens = None
    if world > 1:
        # caption lengths of the GLOBAL batch (same draw on every rank), dealt to the ranks with equal sums: with packed rows a
        # rank's step time follows its row count, and everybody waits at the all-reduce for the rank with the longest captions
        from mvlt_amd.data import deal_balanced
        gl = torch.randint(16, SEQ, (world * PER_GPU_BATCH,), generator=torch.Generator().manual_seed(4242)).tolist()
        lens = [gl[i] for i in deal_balanced(list(range(len(gl))), gl, world)[rank]]
    batch_full = synthetic_batch(PER_GPU_BATCH, SEQ, "cuda", 1234 + rank, with_lengths=True, lengths=lens)
    batch = batch_full[:4]                          # the reference signature: no caption lengths

    for _ in range(args.warmup):
        step(batch)
    every = int(os.environ.get("MVLT_BENCH_SAMPLE", "4"))
    timer = KernelTimer(DOMINANT, every=every)
    native_samples = fam_samples = None
    # ~10 of ~280 launches per step (every bracket costs host time and a barrier packet on the stream); a PRIME period, so
    # the sample does not alias with the 4-6 products per layer (a period of 32 kept hitting the same product of every layer)
    fam_every = 29
    if ops.NATIVE:          # the launches are issued by csrc/host.cpp: it brackets them itself (same method)
        ops.host().timer_begin(0, every, 64 * args.steps)
        ops.host().timer_begin(1, fam_every, 64 * args.steps)
    else:
        ops.GEMM_TIMER = timer
    elapsed, loss = timed_run(step, batch, args.steps, use_dist, dist)
    blocks = block_stats(PER_GPU_BATCH, world)
    ops.GEMM_TIMER = None
    if ops.NATIVE:
        native_samples = ops.host().timer_collect(0)          # [(executed flops, ms)] -- the stream is idle: timed_run synchronised
        fam_samples = ops.host().timer_collect(1)
    loss_value = float(loss.item())

    # EXTRA 1: the same call with config.auto_pack_rows = False -- every zero-padded caption row is computed, i.e.
    # exactly the reference's work (the default plans the packing on the device from the ids themselves)
    dense = None
    if not args.no_extra:
        cfg.auto_pack_rows = False               # (also turns the labelled-rows-first MLM head off: all 80 rows per sample)
        for _ in range(max(2, args.warmup // 2)):
            step(batch)
        e1, l1 = timed_run(step, batch, args.steps, use_dist, dist)
        dense = {"value_dense_rows": round(PER_GPU_BATCH * world * args.steps / e1, 2),
                 "ms_per_step_dense_rows": round(1e3 * e1 / args.steps, 3),
                 "dense_rows_variant": "config.auto_pack_rows=False: every zero-padded caption row computed like the "
                                       "reference (reference-equivalent 132.7 GFLOP/pair executed)",
                 "dense_rows_step_tflops_per_gpu": round(PER_GPU_BATCH * args.steps / e1 * GFLOP_PER_PAIR / 1e3, 2),
                 "dense_rows_step_mfma_frac": round(PER_GPU_BATCH * args.steps / e1 * GFLOP_PER_PAIR / 1e3 / PEAK_BF16_TFLOPS, 4),
                 "dense_rows_loss": round(float(l1.item()), 4)}
        cfg.auto_pack_rows = True
    # EXTRA 2 (opt-in API, not reachable from the reference's unchanged caller): packed BERT rows via the added
    # text_lengths= argument + MLM head on the labelled rows only (config.mlm_max_labels_per_sample = 10)
    extra = None
    if not args.no_extra:
        cfg.mlm_max_labels_per_sample = 10
        for _ in range(max(2, args.warmup // 2)):
            step(batch_full)
        e2, l2 = timed_run(step, batch_full, args.steps, use_dist, dist)
        gpp = packed_gflop_per_pair(batch_full)
        extra = {"value_packed": round(PER_GPU_BATCH * world * args.steps / e2, 2),
                 "ms_per_step_packed": round(1e3 * e2 / args.steps, 3),
                 "packed_variant": "opt-in: forward(..., text_lengths=) packs away padded caption rows; "
                                   "config.mlm_max_labels_per_sample=10 runs the MLM head on labelled rows only "
                                   "(same loss and gradients; tests/test_model_gpu.py::test_packed_rows_*)",
                 "packed_executed_gflop_per_pair": round(gpp, 1),
                 "packed_step_tflops_per_gpu_executed": round(PER_GPU_BATCH * args.steps / e2 * gpp / 1e3, 2),
                 "packed_loss": round(float(l2.item()), 4)}
        cfg.mlm_max_labels_per_sample = None
    if rank == 0:
        pairs = PER_GPU_BATCH * world * args.steps
        value = pairs / elapsed
        kr = timer.result()
        roofline = roofline_entry(fam_samples, DOMINANT_NAME, "family", fam_every)
        roofline_wgrad = roofline_entry(native_samples, WGRAD_NAME, "wgrad_group", every)
        if roofline is None and kr is not None:          # ctypes host path (MVLT_NATIVE_HOST=0): only the grouped kernel is bracketed
            roofline_wgrad = roofline_entry([(kr["tflops"] * 1e12 * kr["avg_us"] * 1e-6, kr["avg_us"] * 1e-3, 0.0)] * kr["launches"],
                                            WGRAD_NAME, "wgrad_group", every)
            roofline = roofline_wgrad
        gpp_exec = packed_gflop_per_pair(batch_full)
        out = {"metric": "image-text pairs/sec pretrain step (Swin-S+BERT, 224px, seq80)", "value": round(value, 2),
               "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": "Pretrain (MLM+ITM) Swin-S + BERT-base, batch=32/GPU, 224x224, seq80, bf16 "
                                      "storage + f32 accumulate/master, fwd+bwd+allreduce+AdamW, random-init weights; "
                                      "reference call signature model(image, caption_masked, caption_label, ITM_label)",
                          "global_batch": PER_GPU_BATCH * world, "seq_len": SEQ, "parallelism": f"dp{world}",
                          "mlm_head_rows": "module default: labelled rows gathered first, their count stays on the device and "
                                           "the head's kernels read it (rows with ignore_index contribute nothing, "
                                           "model.py:410); value_dense_rows also evaluates the head on all 80 rows per sample",
                          "bert_rows": "module default: packing plan computed on the device from the ids/labels (no new "
                                       "argument, no host sync); the zero-padded caption tails are not materialised, "
                                       "loss and gradients equal the dense run (tests/test_model_gpu.py::"
                                       "test_device_planned_packing_*); value_dense_rows = all padded rows computed",
                          "grad_exchange": ("none (1 GPU)" if not use_dist else
                                            f"{'RCCL' if backend == 'nccl' else backend} all-reduce AVG, "
                                            f"{'bf16' if comm == torch.bfloat16 else 'f32'}, {bucket_mb} MiB buckets of the flat "
                                            "gradient arena launched during the backward pass; MLM loss = mean over the labelled "
                                            "tokens of the GLOBAL batch (4-byte label-count all-reduce in the forward pass, "
                                            "ddp.GradReducer.label_sync), so the rank average equals the one-process step on the "
                                            "whole batch"),
                          "loss": round(loss_value, 4)},
               # FLOPs the default path EXECUTES (padded caption rows and unlabelled MLM rows skipped) over the step time; the
               # reference-equivalent 132.7 GFLOP/pair figure is quoted only for the run that executes it (value_dense_rows)
               "executed_gflop_per_pair": round(gpp_exec, 1),
               "step_tflops_per_gpu": round(value / world * gpp_exec / 1e3, 2),
               "step_mfma_frac": round(value / world * gpp_exec / 1e3 / PEAK_BF16_TFLOPS, 4),
               "roofline": roofline, "roofline_wgrad_group": roofline_wgrad, "blocks": blocks,
               # sticky error words of the fused W-MSA kernels' in-launch hand-off (a bounded wait that ran out): must be 0
               "wmsa2_sync_errors": ops.wmsa2_sync_errors()}
        if dense is not None:
            out.update(dense)
        if extra is not None:
            out.update(extra)
        if world == 1 and not args.no_extra:
            out["other_configs"] = other_configs(M)
            if not use_dist:
                out["other_configs"]["one_rank_rccl"] = one_rank_rccl()
                if out["other_configs"]["one_rank_rccl"].get("ms_per_step"):
                    o = out["other_configs"]["one_rank_rccl"]
                    o["plain_step_ms"] = out["ms_per_step"]
                    o["reducer_overhead_ms"] = round((o.get("ms_per_step_without_the_collective") or 0) - out["ms_per_step"], 3) if o.get("ms_per_step_without_the_collective") else None
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        line = json.dumps(out)
        if real_stdout is not None:
            os.write(real_stdout, (line + "\n").encode())
        else:
            print(line, flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
