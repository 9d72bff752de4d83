#!/bin/bash
# Compile-time ablation timing of gg_fwd_patch_k: builds libpai_abl<N>.so variants (container, no GPU) or times
# them on one layer (GPU box):   scripts/abl.sh build "0 4 8 3 15 16"   |   scripts/abl.sh run dec5 "0 4 8 3 15 16"
cd "$(dirname "$0")/.."
P=thesis-pai-reconstruction_amd
if [ "$1" = build ]; then
  for a in $2; do
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -fno-gpu-rdc -mllvm -amdgpu-mfma-vgpr-form=1 -DPATCH_ABL=$a -c $P/csrc/gg_mfma.hip -o /tmp/gg_abl$a.o || exit 1
    objs=$(ls $P/csrc/*.o | grep -v gg_mfma.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libpai_abl$a.so /tmp/gg_abl$a.o $objs || exit 1
    echo built $P/libpai_abl$a.so
  done
else
  for a in $3; do
    echo -n "abl=$a "; PAI_HIP_LIB=$PWD/$P/libpai_abl$a.so timeout 120 python scripts/bench_conv.py $2 2>/dev/null | grep "$2" | head -1
  done
fi
