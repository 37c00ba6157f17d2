"""The loss-head segment of the pre-training forward pass (MLM head on the labelled rows + cross entropy + ITM head,
model.py:399-420 of the reference): how long it takes INSIDE the step without a tracer, against the sum of its kernels.

    python scripts/head_segment.py                       # untraced: HIP-event bracket around the segment, in situ
    rocprofv3 --kernel-trace -d D -o h --output-format csv -- python3 scripts/head_segment.py
    python scripts/head_segment.py D/h_kernel_trace.csv   # reads the trace: kernels between the two marker launches

The segment is delimited by a marker kernel that occurs nowhere else in the step (mvlt_softmax_rows on 4 numbers), launched
right after the encoder returns and right after the loss is formed."""
import csv, os, sys
if len(sys.argv) > 1:
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "softmax_rows" in r["Kernel_Name"]]
    spans, sums, counts = [], [], []
    for a, b in zip(marks[0::2], marks[1::2]):
        seg = rows[a + 1:b]
        if not seg:
            continue
        spans.append((int(rows[b]["Start_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3)
        sums.append(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e3)
        counts.append(len(seg))
    k = len(spans) // 2                     # steady state: second half
    n = len(spans) - k
    print(f"traced run: {len(spans)} segments; steady state: {sum(counts[k:]) / n:.1f} kernels, "
          f"sum of kernel durations {sum(sums[k:]) / n:.1f} us, marker-to-marker span {sum(spans[k:]) / n:.1f} us")
    sys.exit(0)

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvlt_amd as M
from mvlt_amd import ops
from mvlt_amd.train import PretrainStep, synthetic_batch
from mvlt_amd.ddp import seed_coin_flip
torch.manual_seed(0)
cfg = M.MVLBertPretrainConfig(); cfg.ITM_task = True
model = M.MVLBertForPretraining(cfg).cuda().train()
seed_coin_flip(5678)
step = PretrainStep(model)
batch = synthetic_batch(32, 80, "cuda", 1234)[:4]
marker_in = torch.randn(1, 4, device="cuda")
pairs = []
orig = model.MVLBert.forward_autopack


def wrapped(*a, **kw):
    out = orig(*a, **kw)
    ops.softmax_rows(marker_in, 4)
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    pairs.append([e0, None])
    return out


model.MVLBert.forward_autopack = wrapped
orig_fwd = model.forward


def fwd(*a, **kw):
    loss = orig_fwd(*a, **kw)
    ops.softmax_rows(marker_in, 4)
    e1 = torch.cuda.Event(enable_timing=True); e1.record()
    pairs[-1][1] = e1
    return loss


model.forward = fwd
for _ in range(30):
    step(batch)
torch.cuda.synchronize()
t = [a.elapsed_time(b) * 1e3 for a, b in pairs[10:]]
t.sort()
print(f"in situ, HIP events (includes one marker launch and one event record): median {t[len(t) // 2]:.1f} us, "
      f"min {t[0]:.1f}, max {t[-1]:.1f} over {len(t)} steps")
