#!/bin/bash
# timing ablations of gg_wgrad_patch3_k (variants built by: scripts/micro/variants.sh gg_wg3.hip w3a1:-DWG3_ABL=1 w3a4:-DWG3_ABL=4
# w3a8:-DWG3_ABL=8 w3a12:-DWG3_ABL=12 w3a13:-DWG3_ABL=13 w3a5:-DWG3_ABL=5), GPU box.  1 no dW store, 4 no fills, 8 no fragment reads
cd "$GRAFT_REPO_ROOT"
for v in "" w3a1 w3a4 w3a5 w3a8 w3a12 w3a13; do
  if [ -n "$v" ]; then export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/variants/$v; else unset LD_LIBRARY_PATH; fi
  echo "== variant ${v:-full}"
  for L in "$@"; do timeout -k 10 100 scripts/micro/convbench --filter $L --ops w --iters 20 --rounds 3 2>&1 | grep -E "^(dec|enc|D)[0-9]" | cut -c1-110; done
done
