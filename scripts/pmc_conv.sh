#!/bin/bash
# L2 hit/miss counters of the convolution micro-benchmark: scripts/pmc_conv.sh <layer-filter> <outdir>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
f=$1; out=$2
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $out/p1 -- python3 scripts/bench_conv.py $f > /dev/null 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $out/p2 -- python3 scripts/bench_conv.py $f > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for p in ("p1", "p2"):
    files = glob.glob("$out/" + p + "/*/*counter_collection.csv")
    if not files:
        print("no counters", p); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"][:60] + " grid=" + r["Grid_Size"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
    for k, v in acc.items():
        if "gg_" not in k: continue
        print(k, {c: round(x / cnt[(k, c)] / 1e6, 3) for c, x in v.items()}, "(M per launch)")
PY
