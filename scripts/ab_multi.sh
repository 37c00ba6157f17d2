#!/bin/bash
# Interleaved A/B/C... of several environments on ONE box: ab_multi.sh ROUNDS "ENV_A" "ENV_B" ... -- ms/step of bench.py (50 steps)
R=$1; shift
for i in $(seq 1 $R); do
  for arm in "$@"; do
    env $arm python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); b=d['blocks']; print('[$arm]', d['ms_per_step'], 'blocks median', b['ms_per_step_median'], 'min', b['ms_per_step_min'], 'loss', d.get('loss'))"
  done
done
