#!/bin/bash
# timing of gg_fwd_bd_k builds: $1 = variant dir or "", rest = layers
cd "$GRAFT_REPO_ROOT"
v=$1; shift
if [ -n "$v" ]; then export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/variants/$v; fi
for L in "$@"; do
  timeout -k 10 100 scripts/micro/convbench --frag --filter $L --ops fd --iters 20 --rounds 3 --set fwd_bd=0 --set fwd_bd=1 --set fwd_bd=2 2>&1 | grep -E "^(dec|enc|D)[0-9]|MISMATCH|FAIL" | cut -c1-250
done
