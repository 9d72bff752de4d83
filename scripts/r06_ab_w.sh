#!/bin/bash
# wgrad convbench A/B of library variants on one box: scripts/r06_ab_w.sh <tag> variant... ("new" = in-tree)
tag=$1; shift
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
for r in 1 2; do
  for v in "$@"; do
    if [ $v = new ]; then L=thesis-pai-reconstruction_amd; else L=variants/$v; fi
    LD_LIBRARY_PATH=$L timeout -k 10 300 scripts/micro/convbench --ops w --bias --iters 10 --rounds 2 > $out/w_${v}_$r.txt 2>&1 || echo "convbench w $v failed"
    grep total $out/w_${v}_$r.txt | sed "s/^/$v w /"
  done
done
