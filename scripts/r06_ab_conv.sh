#!/bin/bash
# Round-6 A/B of two library builds on ONE box through convbench + the step:  scripts/r06_ab_conv.sh <tag> [variant]
# (variants/<variant>/libpai_hip.so against the in-tree build; interleaved)
tag=$1; var=${2:-base}
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
for r in 1 2; do
  for v in $var new; do
    if [ $v = new ]; then L=thesis-pai-reconstruction_amd; else L=variants/$v; fi
    LD_LIBRARY_PATH=$L timeout -k 10 300 scripts/micro/convbench --ops fd --iters 10 --rounds 2 > $out/fd_${v}_$r.txt 2>&1 || echo "convbench fd $v failed"
    LD_LIBRARY_PATH=$L timeout -k 10 300 scripts/micro/convbench --ops d --bnbwd --iters 10 --rounds 2 > $out/dbn_${v}_$r.txt 2>&1 || echo "convbench dbn $v failed"
    tail -1 $out/fd_${v}_$r.txt | sed "s/^/$v fd /"; grep total $out/dbn_${v}_$r.txt | sed "s/^/$v dbn /"
  done
done
