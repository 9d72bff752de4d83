#!/bin/bash
# package power and clock beside a bare MFMA loop (4 launches of ~2.5 s)
cd "$GRAFT_REPO_ROOT"
scripts/micro/mfma_mix 4000000 bare > gpurun_out/power_bare_mix.log 2>&1 &
pid=$!
sleep 3
for i in 1 2 3 4; do rocm-smi --showpower --showclocks 2>&1 | grep -E "Power \(W\)|sclk" | head -3; sleep 1; done
wait $pid
cat gpurun_out/power_bare_mix.log
