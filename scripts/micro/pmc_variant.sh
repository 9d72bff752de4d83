#!/bin/bash
# Matrix-pipe busy cycles and clock of ONE layer under a variant build of the library (scripts/micro/variants.sh):
#   [CB_EXTRA=--frag] scripts/micro/pmc_variant.sh <variant dir name | -> <layer> <ops> <k=v,...>         (GPU box)
# prints per kernel: launch us, SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM cycles), clock = GRBM_GUI_ACTIVE / 8 / us
v=$1; layer=$2; ops=$3; set=$4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
[ "$v" != "-" ] && export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/variants/$v:$LD_LIBRARY_PATH
out=gpurun_out/pmcv_${v}_${layer}_${set//[=,]/_}
rm -rf $out && mkdir -p $out
B="scripts/micro/convbench $CB_EXTRA --iters 3 --rounds 1 --ops $ops --filter $layer --set $set"
timeout -k 10 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $out/a -- $B > $out/a.log 2>&1 || echo pass a failed
timeout -k 10 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/b -- $B > $out/b.log 2>&1 || echo pass b failed
python3 - $out <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/*/**/*counter_collection.csv", recursive=True):
    dur = {}
    for t in glob.glob(f.rsplit("/", 1)[0] + "/*kernel_trace.csv"):
        for r in csv.DictReader(open(t)):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        if "gg_" not in k:
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] in dur:
            acc[k]["_us_" + r["Counter_Name"]].append(dur[r["Dispatch_Id"]])
for k, c in acc.items():
    v = {n: sum(x) / len(x) for n, x in c.items()}
    us = v.get("_us_GRBM_GUI_ACTIVE", 0)
    cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8
    line = f"{k}: {us:.1f} us, clock {cyc / us / 1e3 if us else 0:.2f} GHz"
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and cyc:
        line += f", mfma busy {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc):.3f} of launch ({v['SQ_VALU_MFMA_BUSY_CYCLES'] / max(v['SQ_INSTS_MFMA'], 1):.1f} cyc/mfma)"
        line += f", wait {v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']:.2f} issue-stall {v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES']:.2f} active {v['SQ_ACTIVE_INST_ANY'] / v['SQ_WAVE_CYCLES']:.2f}"
    print(line)
PY
rm -rf $out/a $out/b
