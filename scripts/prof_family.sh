#!/bin/bash
# kernel statistics of one model family's step (GPU box): scripts/prof_family.sh <tag> <bench.py flags...>
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pf_$tag
rm -rf $out && mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > $out/bench.json 2> $out/stats.err
cp $(ls $out/stats/*/*_kernel_stats.csv | head -1) $out/kernel_stats.csv
rm -rf $out/stats
