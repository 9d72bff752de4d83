#!/bin/bash
# timing ablations of grouped3_k inside the ResNeXt step (variants ga<N> = -DGROUP_ABL=N, scripts/micro/variants.sh gg_group.hip)
# GROUP_ABL bits: 1 no output stores, 2 no LDS fragment reads, 4 no patch fetch
cd "$GRAFT_REPO_ROOT"
for v in "" ga1 ga2 ga4 ga7; do
  if [ -n "$v" ]; then export PAI_HIP_LIB=$GRAFT_REPO_ROOT/variants/$v/libpai_hip.so; else unset PAI_HIP_LIB; fi
  echo "== variant ${v:-full}"
  timeout -k 10 300 python scripts/layer_table.py resnext_unet 2>&1 | grep -E "grouped3_k" | head -4
done
