#!/bin/bash
# kernel trace of the default step, then the in-situ duration of every major kernel class split by its neighbour on the other queue
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${OUT:-insitu}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
PACK=0 MLM_CAP=0 STEPS=10 WARM=5 rocprofv3 --kernel-trace -d $O/step -o s --output-format csv -- python3 $R/scripts/profile_step.py > $O/step.log 2>&1
cd $R
T=$(find $O/step -name "*kernel_trace.csv" | head -1)
python scripts/step_kernels.py $O/step 25 70 > $O/step_kernels.txt
for pat in gemm_glds gemm_kernel attn_bwd ln_bwd gemm_group rowstream gemm8; do python scripts/in_situ_overlap.py $T $pat 40 > $O/in_situ_$pat.txt; done
rm -rf $O/step
