#!/bin/bash
# Stand-alone sweep of the 8-wave engine's tile shape / k-slice count for the Swin stage-0 / 1 weight-gradient groups
for t in 11 12 22; do for s in 8 12 16 20 25 32; do
  echo "== MVLT_G8_TILE=$t MVLT_G8_SPLIT=$s"; MVLT_G8_TILE=$t MVLT_G8_SPLIT=$s ONLY="s0 block,s1 block" python scripts/bench_wgrad_group.py 2>/dev/null
done; done
echo "== default"; ONLY="s0 block,s1 block" python scripts/bench_wgrad_group.py 2>/dev/null
