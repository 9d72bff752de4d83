#!/bin/bash
# SQ / LDS / clock counters of the convolution kernels at the BASELINE configs[1] layer shapes, collected through the
# stand-alone C-ABI micro-benchmark (no Python under the profiler):   scripts/pmc_sq.sh <tag>      (GPU box)
# Separate --pmc passes (8 SQ slots each), --kernel-trace only; summary -> gpurun_out/profiles_<tag>/<tag>_pmc_sq.json
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${tag}_sq
rm -rf $out && mkdir -p $out gpurun_out/profiles_$tag
B="scripts/micro/convbench --iters 3 --rounds 1 --ops fdw $PMC_SQ_ARGS"   # PMC_SQ_ARGS: e.g. "--set wgrad3=0"
pass() {   # name, counters...
  n=$1; shift
  for f in enc2 dec4 dec5 dec6 D1x2 enc4; do
    timeout -k 10 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$n/$f -- $B --filter $f > $out/$n.$f.log 2>&1 || echo "pass $n layer $f failed (see $out/$n.$f.log)"
  done
}
pass p1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA
pass p2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES
pass p3 GRBM_GUI_ACTIVE
python3 scripts/pmc_sq.py $out gpurun_out/profiles_$tag/${tag}_pmc_sq.json
rm -rf $out/p1 $out/p2 $out/p3
