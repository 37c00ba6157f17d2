#!/bin/bash
# Interleaved A/B of two environments on ONE box: ab_bench.sh "ENV_A" "ENV_B" [rounds] -- prints ms/step medians of bench.py blocks
A="$1"; B="$2"; R=${3:-3}
for i in $(seq 1 $R); do
  for arm in "$A" "$B"; do
    env $arm python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); b=d['blocks']; print('[$arm]', d['ms_per_step'], 'blocks median', b['ms_per_step_median'], 'min', b['ms_per_step_min'], 'max', b['ms_per_step_max'])"
  done
done
