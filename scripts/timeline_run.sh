#!/bin/bash
# kernel trace of a step as a timeline (GPU box): scripts/timeline_run.sh [bench.py flags...]
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tl && mkdir -p gpurun_out/tl
timeout -k 10 300 python3 bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > /dev/null 2> gpurun_out/tl/prewarm.err
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl/tr -- python3 bench.py "$@" --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events > gpurun_out/tl/bench.json 2> gpurun_out/tl/tr.err
python3 scripts/timeline.py $(ls gpurun_out/tl/tr/*/*_kernel_trace.csv | head -1) 4 --seq > gpurun_out/tl/timeline.txt
cp $(ls gpurun_out/tl/tr/*/*_kernel_trace.csv | head -1) gpurun_out/tl/kernel_trace.csv && rm -rf gpurun_out/tl/tr
