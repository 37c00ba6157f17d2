#!/bin/bash
# GPU-box helper for the fused W-MSA v2 work: parity test, stand-alone timing, then the phase timeline from a -DW2_TRACE rebuild
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "wmsa2" > gpurun_out/t_wmsa2.log 2>&1
rc=$?; tail -4 gpurun_out/t_wmsa2.log
[ $rc -ne 0 ] && exit $rc
STAGE=${STAGE:-2} ONLY=fused timeout -k 10 300 python scripts/bench_wmsa.py > gpurun_out/wmsa2_bench.txt 2>&1 || exit 1
grep -v "head-split" gpurun_out/wmsa2_bench.txt
cd medical-vision-langauge-transformer_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-unused-result -DW2_TRACE -c wmsa2.hip -o wmsa2.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmvlt_hip.so gemm.o gemm8.o rowstream.o norm.o attn.o misc.o wmsa.o wmsa2.o && cd ../.. || exit 1
STAGE=${STAGE:-2} timeout -k 10 120 python scripts/wmsa2_trace.py > gpurun_out/wmsa2_trace.txt 2>&1 && STAGE=${STAGE:-2} SAVE=1 timeout -k 10 120 python scripts/wmsa2_trace.py >> gpurun_out/wmsa2_trace.txt 2>&1
grep -v amdgpu.ids gpurun_out/wmsa2_trace.txt
# ablation builds (timing only: their results are wrong): ABL="-DW2_NO_WLOAD -DW2_NO_AREAD ..." one build per flag
for f in $ABL; do
  (cd medical-vision-langauge-transformer_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-unused-result -DW2_TRACE $f -c wmsa2.hip -o wmsa2.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmvlt_hip.so gemm.o gemm8.o rowstream.o norm.o attn.o misc.o wmsa.o wmsa2.o) || exit 1
  echo "=== ablation $f" >> gpurun_out/wmsa2_trace.txt; echo "=== ablation $f"
  STAGE=${STAGE:-2} timeout -k 10 120 python scripts/wmsa2_trace.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/wmsa2_trace.txt
done
# restore the product build (the diagnostic objects above are newer than the source: make alone would keep them)
rm -f medical-vision-langauge-transformer_amd/csrc/wmsa2.o && make -C medical-vision-langauge-transformer_amd/csrc -j8 > /dev/null 2>&1 && echo "product build restored"
