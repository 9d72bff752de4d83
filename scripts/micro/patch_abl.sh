#!/bin/bash
# timing ablations of gg_fwd_patch_k (variants pa<N> = -DPATCH_ABL=N built by scripts/micro/variants.sh gg_mfma.hip), GPU box
# usage: patch_abl.sh <ops> <layer>...     PATCH_ABL bits: 1 no weight fill, 2 no patch fill, 4 no MFMA, 8 no fragment reads, 16 no epilogue
cd "$GRAFT_REPO_ROOT"
O=$1; shift
for v in "" pa1 pa2 pa3 pa8 pa11 pa16 pa27; do
  if [ -n "$v" ]; then export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/variants/$v; else unset LD_LIBRARY_PATH; fi
  echo "== variant ${v:-full}"
  for L in "$@"; do timeout -k 10 100 scripts/micro/convbench --filter $L --ops $O --iters 20 --rounds 3 2>&1 | grep -E "^(dec|enc|D)[0-9]" | cut -c1-125; done
done
