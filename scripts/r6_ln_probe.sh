#!/bin/bash
# round 6: where does LayerNorm backward lose its time inside the step?  (1) kernel trace of the default step with the
# concurrency kept, split by neighbour (scripts/in_situ_overlap.py); (2) counter passes over the same launches (rocprofv3
# serialises dispatches under --pmc: these are the stand-alone numbers of the in-step shapes).
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${OUT:-ln6}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
PACK=0 MLM_CAP=0 STEPS=10 WARM=5 rocprofv3 --kernel-trace -d $O/step -o s --output-format csv -- python3 $R/scripts/profile_step.py > $O/step.log 2>&1
cd $R
T=$(find $O/step -name "*kernel_trace.csv" | head -1)
python scripts/step_kernels.py $O/step 25 70 > $O/step_kernels.txt
python scripts/in_situ_overlap.py $T ln_bwd 40 > $O/ln_bwd_in_situ.txt
python scripts/in_situ_overlap.py $T ln_fwd 40 > $O/ln_fwd_in_situ.txt
python scripts/trace_window.py $T "ln_bwd_kernelIDF16bLi64ELi3ELb0" -20 6 40 > $O/window_bert.txt
cd /tmp
if [ -z "$NOPMC" ]; then
python3 $R/scripts/pmc.py --out $O/pmc --match ln_bwd --passes sq1,sq2 -- python3 $R/bench.py --no-extra --no-cpu-baseline --steps 3 --warmup 2 > $O/ln_bwd_pmc.txt 2>$O/pmc.err
fi
rm -rf $O/step $O/pmc
echo probe done
