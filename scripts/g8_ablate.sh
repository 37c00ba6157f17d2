#!/bin/bash
# gemm8 ablation: time of the K loop without MFMA / without LDS-DMA (outputs are wrong in those builds)
cd medical-vision-langauge-transformer_amd/csrc
for flags in "" "-DG8_NO_MMA" "-DG8_NO_GLDS" "-DG8_NO_MMA -DG8_NO_GLDS"; do
  touch gemm8.hip; make EXTRA="$flags" >/dev/null 2>&1
  echo "== build [$flags]"
  (cd ../.. && MVLT_G8=1 python - <<'PY'
import os, sys, torch
sys.path.insert(0, ".")
from mvlt_amd import ops
dt = torch.bfloat16
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
A = (torch.randn(4096, 4096, device="cuda") * .5).to(dt); B = (torch.randn(4096, 4096, device="cuda") * .5).to(dt); o = torch.empty(4096, 4096, dtype=dt, device="cuda")
for tile in ("22", "12"):
    os.environ["MVLT_G8_TILE"] = tile
    print(f"  4096^3 rr tile {tile}: {t(lambda: ops.gemm(A, B, out=o)):7.1f} us   rk: {t(lambda: ops.gemm(A, B, b_kmajor=True, out=o)):7.1f} us")
ws = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
items = [((torch.randn(3090, no, device="cuda") * .5).to(dt), (torch.randn(3090, ni, device="cuda") * .5).to(dt), torch.zeros(no, ni, device="cuda"), torch.zeros(no, device="cuda")) for no, ni in ws]
for tile in ("22", "12", "11"):
    os.environ["MVLT_G8_TILE"] = tile
    print(f"  bert wgrad group tile {tile}: {t(lambda: ops.wgrad_group(items)):7.1f} us")
PY
)
done
touch gemm8.hip; make >/dev/null 2>&1
