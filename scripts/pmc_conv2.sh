#!/bin/bash
# SQ counters (matrix-pipe busy share, waits) of the convolution micro-benchmark: scripts/pmc_conv2.sh <layer-filter> <outdir>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
f=$1; out=$2
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $out/p1 -- python3 scripts/bench_conv.py $f > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $out/p2 -- python3 scripts/bench_conv.py $f > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $out/p3 -- python3 scripts/bench_conv.py $f > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for p in ("p1", "p2", "p3"):
    files = glob.glob("$out/" + p + "/*/*counter_collection.csv")
    if not files:
        print("no counters", p); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"][:44] + " grid=" + r["Grid_Size"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
    for k, v in acc.items():
        if "gg_fwd" not in k: continue
        print(k, {c: round(x / cnt[(k, c)] / 1e6, 3) for c, x in v.items()})
PY
