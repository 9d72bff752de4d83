#!/bin/bash
# Builds libpai_hip.so variants that differ in the compile-time switches of ONE translation unit (container, no GPU):
#   scripts/micro/variants.sh gg_p2.hip name1:-DP2_ABL=1 name2:"-DP2_ABL=2 -DP2_SETPRIO=1" ...
# -> variants/<name>/libpai_hip.so; run with  LD_LIBRARY_PATH=variants/<name> scripts/micro/convbench ...
cd "$(dirname "$0")/../.."
P=thesis-pai-reconstruction_amd
src=$1; shift
base=${src%.hip}
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  mkdir -p variants/$name
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -fno-gpu-rdc -mllvm -amdgpu-mfma-vgpr-form=1 $flags -c $P/csrc/$src -o variants/$name/$base.o || exit 1
  objs=$(ls $P/csrc/*.o | grep -v "/$base.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/$name/libpai_hip.so variants/$name/$base.o $objs -ldl || exit 1
  echo built variants/$name/libpai_hip.so "($flags)"
done
