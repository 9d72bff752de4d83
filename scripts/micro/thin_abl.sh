#!/bin/bash
# timing ablations of thin_fwd_k, warm and cold (variants ta<N> = -DTHIN_ABL=N built by scripts/micro/variants.sh gg_thin.hip), GPU box
# THIN_ABL bits: 1 no gathers, 2 no stores
cd "$GRAFT_REPO_ROOT"
for v in "" ta1 ta2 ta3; do
  if [ -n "$v" ]; then export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/variants/$v; else unset LD_LIBRARY_PATH; fi
  for c in "" "--cold"; do
    echo "== variant ${v:-full} ${c:-warm}"
    timeout -k 10 100 scripts/micro/convbench --filter thin_ --ops fdw --iters 10 --rounds 3 $c 2>&1 | grep -E "^thin" | cut -c1-140
  done
done
