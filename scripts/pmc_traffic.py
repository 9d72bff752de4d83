"""Per-kernel HBM traffic from two separate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of bench.py.

    python scripts/pmc_traffic.py <fetch_dir> <write_dir> <out_json> [<raw_json>]

bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE / WRITE_SIZE are in KiB, and on
gfx950 FETCH_SIZE tallies a wide coalesced read at half its bytes (MI355X_MICROARCH.md, HBM section).
Keys are kernel symbols as `pai_conv_kernel_name` / rocprofv3 print them ("void " and the argument
list stripped), so bench.py can look its dominant kernel up directly.
"""
import collections
import csv
import glob
import json
import re
import sys


def symbol(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    depth = 0
    for i, ch in enumerate(name):       # cut the argument list: first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name


def collect(d, counter):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    acc, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != counter:
            continue
        k = symbol(r["Kernel_Name"])
        acc[k] += float(r["Counter_Value"])
        cnt[k] += 1
    return {k: (acc[k] / cnt[k], cnt[k]) for k in acc}


def main():
    fetch = collect(sys.argv[1], "FETCH_SIZE")
    write = collect(sys.argv[2], "WRITE_SIZE")
    out = {"_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, with --kernel-trace) of "
                      "`python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-events`; bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 "
                      "per launch (gfx950: FETCH_SIZE reports half of a wide coalesced read, MI355X_MICROARCH.md HBM section)"}
    for k in sorted(fetch, key=lambda k: -fetch[k][0] * fetch[k][1]):
        f, n = fetch[k]
        w = write.get(k, (0.0, 0))[0]
        out[k] = {"launches_sampled": n, "fetch_kib_avg": round(f, 1), "write_kib_avg": round(w, 1),
                  "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in list(out.items())[1:16]:
        print(f"{v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch  x{v['launches_sampled']:<5} {k}")


if __name__ == "__main__":
    main()
