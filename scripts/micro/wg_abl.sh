#!/bin/bash
# timing ablations of gg_wgrad_patch_k (variants built by scripts/micro/variants.sh gg_mfma.hip wga...), GPU box
cd "$GRAFT_REPO_ROOT"
for v in "" wga1 wga4 wga8 wga12 wga13; do
  if [ -n "$v" ]; then export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/variants/$v; else unset LD_LIBRARY_PATH; fi
  echo "== variant ${v:-full}"
  for L in "$@"; do timeout -k 10 100 scripts/micro/convbench --filter $L --ops w --iters 20 --rounds 3 2>&1 | grep -E "^(dec|enc|D)[0-9]" | cut -c1-110; done
done
