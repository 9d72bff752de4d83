#!/bin/bash
# Interleaved step-level A/B on ONE box: scripts/ab_step.sh ROUNDS label[:lib=PATH][:set=name=v,...][:env=NAME=V] ...
# prints ms_per_step of `bench.py --no-cpu-baseline --no-kernel-events` per label and round
cd "$GRAFT_REPO_ROOT"
rounds=$1; shift
for i in $(seq 1 $rounds); do
  for spec in "$@"; do
    label=${spec%%:*}; lib=""; set=""
    IFS=':' read -ra parts <<< "$spec"
    envs=""
    for p in "${parts[@]:1}"; do case $p in lib=*) lib=${p#lib=};; set=*) set=${p#set=};; env=*) envs="$envs ${p#env=}";; esac; done
    if [ -n "$lib" ]; then export PAI_HIP_LIB=$GRAFT_REPO_ROOT/$lib; else unset PAI_HIP_LIB; fi
    ms=$(env $envs timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-events ${set:+--set $set} 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$label $ms"
  done
done
