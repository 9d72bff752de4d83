#!/bin/bash
# convbench over library variants on ONE box:  scripts/r06_variants.sh <tag> "<convbench args>" variant...   ("new" = in-tree)
tag=$1; args=$2; shift 2
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
for r in 1 2; do
  for v in "$@"; do
    if [ $v = new ]; then L=thesis-pai-reconstruction_amd; else L=variants/$v; fi
    LD_LIBRARY_PATH=$L timeout -k 10 300 scripts/micro/convbench $args > $out/${v}_$r.txt 2>&1 || echo "convbench $v failed"
    grep " med " $out/${v}_$r.txt | awk -v v=$v '{printf "%s %s %s %s | ", v, $1, $2, $7} END {print ""}'
  done
done
