#!/bin/bash
# package power and clock beside the training step (bench.py, 1500 steps)
cd "$GRAFT_REPO_ROOT"
python3 bench.py --steps 5000 --warmup 20 --no-cpu-baseline --no-kernel-events > gpurun_out/power_step_bench.json 2>/dev/null &
pid=$!
sleep 16
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks 2>&1 | grep -E "Power \(W\)|sclk" | head -3; sleep 1; done
wait $pid
python3 -c "import json; d=json.loads(open('gpurun_out/power_step_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
