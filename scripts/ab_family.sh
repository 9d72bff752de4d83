#!/bin/bash
# Interleaved A/B of another model family's step on ONE box: scripts/ab_family.sh ROUNDS "<bench flags>" label[:lib=PATH][:set=k=v][:env=NAME=V] ...
cd "$GRAFT_REPO_ROOT"
rounds=$1; flags=$2; shift; shift
for i in $(seq 1 $rounds); do
  for spec in "$@"; do
    label=${spec%%:*}; lib=""; set=""; envs=""
    IFS=':' read -ra parts <<< "$spec"
    for p in "${parts[@]:1}"; do case $p in lib=*) lib=${p#lib=};; set=*) set=${p#set=};; env=*) envs="$envs ${p#env=}";; esac; done
    if [ -n "$lib" ]; then export PAI_HIP_LIB=$GRAFT_REPO_ROOT/$lib; else unset PAI_HIP_LIB; fi
    ms=$(env $envs timeout -k 10 300 python bench.py $flags --no-cpu-baseline --no-kernel-events ${set:+--set $set} 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$label $ms"
  done
done
