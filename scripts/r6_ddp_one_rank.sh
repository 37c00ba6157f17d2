#!/bin/bash
# Round 6, VERDICT r5 item 7: what ONE GPU can show of the data-parallel step (interleaved arms of bench.py, 50 steps each):
#   plain            no reducer
#   null             reducer, no collective (MVLT_DDP_NULL_COLLECTIVE=1): the reducer's own cost
#   rccl             reducer + RCCL on one rank (a one-rank all-reduce is a device copy of the arena at HBM speed)
#   rccl-deferred    the same, end-of-backward wait deferred (MVLT_DDP_DEFER_WAIT=1): rccl - this = EXPOSED tail of the exchange
#   ring b,i         rehearsal of world size 8: every bucket copied twice by b persistent workgroups with i loads in flight per
#                    thread on a third stream (MVLT_DDP_REHEARSE): a collective that stays busy for milliseconds at a few
#                    hundred GB/s beside the backward pass, with and without the tapered tail buckets
R=${1:-2}
F="MVLT_FORCE_DDP=1"
bash scripts/ab_multi.sh $R "MVLT_X=plain" "$F MVLT_DDP_NULL_COLLECTIVE=1" "$F" "$F MVLT_DDP_DEFER_WAIT=1" \
  "$F MVLT_DDP_REHEARSE=32,4" "$F MVLT_DDP_REHEARSE=32,4 MVLT_DDP_DEFER_WAIT=1" "$F MVLT_DDP_REHEARSE=32,4 MVLT_DDP_TAPER_MB=0" \
  "$F MVLT_DDP_REHEARSE=64,4" "$F MVLT_DDP_REHEARSE=64,4 MVLT_DDP_DEFER_WAIT=1" "$F MVLT_DDP_REHEARSE=64,4 MVLT_DDP_TAPER_MB=0"
python - <<'PY'
# stand-alone rate of the rehearsal copy (GB/s of bytes copied = read + written / 2)
import torch, sys, os
sys.path.insert(0, os.getcwd())
import mvlt_amd
from mvlt_amd import ops
a = torch.empty(64 << 20, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
for blocks, infl in ((32, 2), (32, 4), (64, 4), (128, 4)):
    for _ in range(2): ops.debug_stream_copy(b, a, blocks, infl)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.debug_stream_copy(b, a, blocks, infl)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"rehearsal copy {blocks} workgroups x {infl} loads in flight: {a.numel() * 4 / ms / 1e6:.0f} GB/s copied (256 MiB in {ms:.2f} ms)")
PY
