#!/bin/bash
# thin weight gradients, cold, against the workgroup cap (PAI_TW_BLOCKS; 0 = default 1024 / channel groups)
cd "$GRAFT_REPO_ROOT"
for b in 0 512 1536 2048 2560 4096; do
  echo "== PAI_TW_BLOCKS=$b"
  PAI_TW_BLOCKS=$b timeout -k 10 100 scripts/micro/convbench --filter thin_ --ops w --iters 10 --rounds 3 --cold 2>&1 | grep -E "^thin" | cut -c1-110
done
