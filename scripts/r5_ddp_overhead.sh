#!/bin/bash
# One rank over RCCL (MVLT_FORCE_DDP=1): what the gradient reducer costs the step and where.
# Interleaved arms on one box, ms per step of bench.py (50 steps).
scripts/ab_multi.sh ${1:-2} "MVLT_X=none" "MVLT_FORCE_DDP=1" "MVLT_FORCE_DDP=1 MVLT_DDP_NULL_COLLECTIVE=1" \
   "MVLT_FORCE_DDP=1 MVLT_DDP_BUCKET_MB=256" "MVLT_FORCE_DDP=1 MVLT_DDP_BUCKET_MB=1024" "MVLT_FORCE_DDP=1 MVLT_DDP_FORK=1" \
   "MVLT_FORCE_DDP=1 MVLT_DDP_FORK=1 MVLT_DDP_NULL_COLLECTIVE=1"
