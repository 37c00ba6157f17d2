#!/bin/bash
# Stand-alone times of the step's tile-kernel products under each forced number of column groups of the tile order
# (MVLT_XCD_CS = 1 off, 2 / 4 / 8 forced where it divides the column tiles, 0 = the cost model's choice).
B="3150,2304,768,0,0,b;3150,768,768,0,0,br;3150,3072,768,0,0,g;3150,768,3072,0,0,br;3150,3072,768,0,1,a;3150,768,3072,0,1;3150,768,768,0,1;3150,768,2304,0,1"
S2="6272,1536,384,0,0,g;6272,384,1536,0,0,br;6272,384,1536,0,1;6272,1536,384,0,1,a;6272,384,1152,0,1;6272,384,384,0,1"
S3="1568,3072,768,0,0,g;1568,768,3072,0,0,br;1568,3072,768,0,1,a;1568,768,3072,0,1;1568,2304,768,0,0,b"
S1="25088,768,192,0,0,g;25088,192,768,0,0,br;25088,768,192,0,1,a;25088,192,768,0,1"
D="4192,3072,768,0,0,g;4192,768,3072,0,0,br;4192,2304,768,0,0,b"
for cs in 1 0 2 4 8; do
  echo "== MVLT_XCD_CS=$cs"
  MVLT_XCD_CS=$cs SHAPES="$B;$S2;$S3;$S1;$D" python scripts/bench_gemm_shape.py 2>/dev/null
done
