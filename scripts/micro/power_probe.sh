#!/bin/bash
# package power and clocks while one convolution layer runs back to back (GPU box): $1 = layer, $2 = ops, $3.. = convbench flags
cd "$GRAFT_REPO_ROOT"
L=$1; O=$2; shift 2
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -E "Power|sclk|mclk|Max" | head -8
scripts/micro/convbench --filter $L --ops $O --iters 40000 --rounds 2 "$@" > gpurun_out/power_cb.log 2>&1 &
pid=$!
sleep 6
for i in 1 2 3 4; do rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk" | head -3; sleep 1; done
wait $pid
grep -E "^(dec|enc|D)[0-9]" gpurun_out/power_cb.log | cut -c1-200
