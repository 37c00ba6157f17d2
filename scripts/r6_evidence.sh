#!/bin/bash
# Final evidence of round 6 on one box: counter passes over the step's GEMM kernels (family traffic / MFMA busy), the traced bench
# with its kernel summary and the live-vs-trace check, the per-step kernel table, the decode kernel summary, then the un-profiled
# bench line.  Everything lands in gpurun_out/ev6; the summaries are copied to profiles/r6_* afterwards.
# PART=A: counter passes + traces of the step; PART=B: decode, W-MSA counters, the un-profiled bench line (two calls: a gpurun call is
# limited to 20 minutes)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ev6; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
if [ "${PART:-A}" = "A" ]; then
python3 $R/scripts/pmc.py --out $O/pmc --match gemm --passes sq1,fetch,write \
  --family "family=gemm_kernel|gemm_glds_kernel|rowstream_kernel|gemm8_kernelILi.ELi.ELb0|gemm8_kernel<., ., false" --family "wgrad_group=gemm_group_kernel|gemm_group_glds_kernel" \
  --json $O/r6_dominant_kernel_traffic.json -- python3 $R/bench.py --no-extra --no-cpu-baseline --steps 4 --warmup 2 > $O/r6_step_gemm_pmc.txt 2>$O/pmc.err
echo pmc done
python3 $R/scripts/pmc.py --out $O/pmc_rs --match rowstream --passes sq1,fetch,write -- python3 $R/bench.py --no-extra --no-cpu-baseline --steps 4 --warmup 2 > $O/r6_step_rowstream_pmc.txt 2>$O/pmc_rs.err
echo pmc rowstream done
rocprofv3 --kernel-trace --stats -d $O/trace -o b --output-format csv -- python3 $R/bench.py --no-extra --no-cpu-baseline --steps 30 --warmup 10 > $O/traced_bench.json 2>$O/trace.err
echo trace done
cd $R
python scripts/roofline_vs_trace.py $(find $O/trace -name "*kernel_trace.csv" | head -1) $O/traced_bench.json 10 30 > $O/r6_roofline_vs_trace.txt 2>&1 || true
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/r6_bench_kernel_stats.csv
cd /tmp
PACK=0 MLM_CAP=0 STEPS=10 WARM=5 rocprofv3 --kernel-trace -d $O/step -o s --output-format csv -- python3 $R/scripts/profile_step.py > $O/step.log 2>&1
cd $R
python scripts/step_kernels.py $O/step 25 70 > $O/r6_step_kernels.txt
PACK=0 MLM_CAP=0 STAMPS=1 STEPS=3 WARM=10 python scripts/profile_step.py 2>&1 | grep -v "^[WE]2026\|amdgpu.ids" > $O/r6_step_phases.txt
T=$(find $O/step -name "*kernel_trace.csv" | head -1)
python scripts/in_situ_overlap.py $T ln_bwd 40 > $O/r6_ln_bwd_in_situ_final.txt || true
python scripts/in_situ_overlap.py $T wmsa2 40 > $O/r6_wmsa2_in_situ.txt || true
rm -rf $O/trace $O/step $O/pmc $O/pmc_rs
echo part A done
exit 0
fi
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/dec -o d --output-format csv -- python3 $R/scripts/bench_decode.py > $O/decode.log 2>&1 || true
cp $(find $O/dec -name "*kernel_stats.csv" | head -1) $O/r6_decode_kernel_stats.csv || true
cd /tmp
for st in 0 1 2; do
  STAGE=$st python3 $R/scripts/pmc.py --out $O/pmc_w$st --match wmsa2 --passes sq1,sq2 -- python3 $R/scripts/bench_wmsa2.py > $O/r6_wmsa2_pmc_s$st.txt 2>$O/pmc_w$st.err || true
done
STAGE=2 SAVE=1 python3 $R/scripts/pmc.py --out $O/pmc_w2s --match wmsa2 --passes sq1 -- python3 $R/scripts/bench_wmsa2.py > $O/r6_wmsa2_pmc_s2_save.txt 2>$O/pmc_w2s.err || true
cd $R
ONLY=fused python scripts/bench_wmsa.py > $O/r6_wmsa2_bench.txt 2>&1 || true
rm -rf $O/dec $O/pmc_w0 $O/pmc_w1 $O/pmc_w2 $O/pmc_w2s
python bench.py > $O/r6_bench.json 2> $O/bench.err
tail -c 600 $O/r6_bench.json
