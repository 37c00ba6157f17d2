#!/bin/bash
# interleaved A/B of two library builds on one box: ab_lib.sh ROUNDS libA.so libB.so
R=$1; A=$2; B=$3
for i in $(seq 1 $R); do
  for L in $A $B; do
    cp $L medical-vision-langauge-transformer_amd/libmvlt_hip.so
    python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); b=d['blocks']; print('[$L]', d['ms_per_step'], 'blocks median', b['ms_per_step_median'], 'min', b['ms_per_step_min'])"
  done
done
