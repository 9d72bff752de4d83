#!/bin/bash
# A/B two builds of libpai_hip.so on the same GPU box: scripts/ab.sh [bench args]
# base = thesis-pai-reconstruction_amd/libpai_hip_base.so (copy of an earlier build with the same ABI),
# new = the in-tree build.  bench.py ramps the clocks itself before timing.
cd "$(dirname "$0")/.."
for i in 1 2; do
  if [ -z "$AB_NO_BASE" ]; then
    PAI_HIP_LIB=$PWD/thesis-pai-reconstruction_amd/libpai_hip_base.so timeout 300 python bench.py --no-cpu-baseline "$@" | python scripts/bench_line.py base
  fi
  timeout 300 python bench.py --no-cpu-baseline "$@" | python scripts/bench_line.py new
done
