#!/bin/bash
# builds the ablation variants of the row-streaming kernel on the GPU box and times them (timing only: outputs are wrong)
set -e
for v in "" "-DRS_ABL_NOGELU" "-DRS_ABL_NOSTORE" "-DRS_ABL_NOMMA" "-DRS_ABL_NOGELU -DRS_ABL_NOSTORE" "-DRS_ABL_NOGELU -DRS_ABL_NOSTORE -DRS_ABL_NOMMA"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value $v -o /tmp/rs_abl scripts/rowstream_ablate.hip 2>/dev/null
  for c in 0 1 2; do echo -n "[${v:-full}] "; timeout -k 5 60 /tmp/rs_abl $c; done
done
