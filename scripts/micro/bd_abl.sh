#!/bin/bash
# timing ablations of gg_fwd_bd_k (variants built by scripts/micro/variants.sh gg_bd.hip ...), GPU box
cd "$GRAFT_REPO_ROOT"
for v in "" bdw bdr bdp bdwr bdnone; do
  if [ -n "$v" ]; then export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/variants/$v; else unset LD_LIBRARY_PATH; fi
  echo "== variant ${v:-full}"
  timeout -k 10 100 scripts/micro/convbench --frag --filter ${1:-dec4} --ops ${2:-fd} --iters 20 --rounds 3 --set fwd_bd=1 2>&1 | grep -E "^(dec|enc|D)[0-9]"
done
