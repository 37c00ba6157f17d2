#!/bin/bash
# Final evidence of round 5 on one box: counter passes over the step's GEMM kernels (family traffic / MFMA busy), the traced bench
# with its kernel summary and the live-vs-trace check, the per-step kernel table, the decode kernel summary, then the un-profiled
# bench line.  Everything lands in gpurun_out/ev5; the summaries are copied to profiles/r5_* afterwards.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ev5; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python3 $R/scripts/pmc.py --out $O/pmc --match gemm --passes sq1,fetch,write \
  --family "family=gemm_kernel|gemm_glds_kernel|rowstream_kernel|gemm8_kernelILi.ELi.ELb0|gemm8_kernel<., ., false" --family "wgrad_group=gemm_group_kernel" \
  --json $O/r5_dominant_kernel_traffic.json -- python3 $R/bench.py --no-extra --no-cpu-baseline --steps 4 --warmup 2 > $O/r5_step_gemm_pmc.txt 2>$O/pmc.err
echo pmc done
python3 $R/scripts/pmc.py --out $O/pmc_rs --match rowstream --passes sq1,fetch,write -- python3 $R/bench.py --no-extra --no-cpu-baseline --steps 4 --warmup 2 > $O/r5_step_rowstream_pmc.txt 2>$O/pmc_rs.err
echo pmc rowstream done
rocprofv3 --kernel-trace --stats -d $O/trace -o b --output-format csv -- python3 $R/bench.py --no-extra --no-cpu-baseline --steps 30 --warmup 10 > $O/traced_bench.json 2>$O/trace.err
echo trace done
cd $R
python scripts/roofline_vs_trace.py $(find $O/trace -name "*kernel_trace.csv" | head -1) $O/traced_bench.json 10 30 > $O/r5_roofline_vs_trace.txt 2>&1 || true
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/r5_bench_kernel_stats.csv
cd /tmp
PACK=0 MLM_CAP=0 STEPS=10 WARM=5 rocprofv3 --kernel-trace -d $O/step -o s --output-format csv -- python3 $R/scripts/profile_step.py > $O/step.log 2>&1
cd $R
python scripts/step_kernels.py $O/step 25 70 > $O/r5_step_kernels.txt
PACK=0 MLM_CAP=0 STAMPS=1 STEPS=3 WARM=10 python scripts/profile_step.py 2>&1 | grep -v "^[WE]2026\|amdgpu.ids" > $O/r5_step_phases.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/dec -o d --output-format csv -- python3 $R/scripts/bench_decode.py > $O/decode.log 2>&1 || true
cp $(find $O/dec -name "*kernel_stats.csv" | head -1) $O/r5_decode_kernel_stats.csv || true
cd $R
rm -rf $O/trace $O/step $O/dec $O/pmc $O/pmc_rs
python bench.py > $O/r5_bench.json 2> $O/bench.err
tail -c 600 $O/r5_bench.json
