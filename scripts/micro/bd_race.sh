#!/bin/bash
# repeated exactness runs of gg_fwd_bd_k on grids of several rounds (batch 128); $1 = variant dir or ""
cd "$GRAFT_REPO_ROOT"
if [ -n "$1" ]; then export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/variants/$1; fi
for r in 1 2 3; do for L in dec5 enc2 D2x2; do
  timeout -k 10 100 scripts/micro/convbench --frag --filter $L --ops fd --batch 128 --iters 3 --rounds 1 --set fwd_bd=0 --set fwd_bd=1 --set fwd_bd=2 2>&1 | grep -E "MISMATCH|agree|FAIL"
done; done
