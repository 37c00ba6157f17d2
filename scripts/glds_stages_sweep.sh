#!/bin/bash
# 64 x 64 LDS-DMA tiles with two LDS stages (one K-tile in flight per workgroup) against three (two in flight, counted vmcnt), stand-alone
SH="3150,768,3072,0,0,br;3150,768,768,0,0,br;6272,384,1536,0,0,br;6272,384,384,0,0,br;3150,768,3072,0,1;3150,768,2304,0,1;3150,768,768,0,1;6272,384,1536,0,1;6272,384,1152,0,1;1568,768,3072,0,0,br;4192,768,3072,0,0,br"
for i in 1 2; do for m in 0 2; do echo "== MVLT_GLDS_STAGES=$m"; MVLT_GLDS_STAGES=$m SHAPES="$SH" python scripts/bench_gemm_shape.py 2>/dev/null; done; done
