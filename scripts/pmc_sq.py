"""Reduces the rocprofv3 --pmc passes of scripts/pmc_sq.sh to one JSON: per kernel symbol (and grid size, i.e. layer)
the average counter values per launch, the launch duration from the kernel trace of the same pass, and the derived
figures the CDNA4 guide defines:

  mfma_busy_frac   = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)   (matrix-pipe busy cycles per SIMD-cycle the CU
                     had a wave on it; SQ_VALU_MFMA_BUSY_CYCLES is summed over the 4 SIMDs of every CU)
  mfma_busy_vs_sq  = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES            (the ratio VERDICT r01 asks for, as the counters come)
  cycles_per_mfma  = SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA             (16 for v_mfma_f32_16x16x32_bf16: sanity check)
  wait_frac / issue_stall_frac / active_frac = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES
  lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  clock_ghz        = GRBM_GUI_ACTIVE / 8 / duration                        (guide: 'DVFS give-back')
  mfma_busy_of_launch = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * duration * clock): share of the launch during which a
                     SIMD's matrix pipe is busy -- the figure to hold against the 2.5 PFLOP/s peak (x clock / 2.4 GHz)

    python scripts/pmc_sq.py <dir of passes> <out.json>
"""
import collections
import csv
import glob
import json
import re
import sys


def symbol(name):
    name = re.sub(r"^void ", "", name)
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name


def main(root, out_path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(root + "/p*/*/**/*counter_collection.csv", recursive=True):
        layer = f.split("/")[-3]            # <root>/<pass>/<layer>/<host>/<pid>_counter_collection.csv
        dur = {}
        for t in glob.glob(f.rsplit("/", 1)[0] + "/*kernel_trace.csv"):
            for r in csv.DictReader(open(t)):
                dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
        for r in csv.DictReader(open(f)):
            k = symbol(r["Kernel_Name"])
            if not (k.startswith("gg_") or k.startswith("thin_") or k.startswith("splitk")):
                continue
            key = (layer, k, r["Grid_Size"])
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Dispatch_Id"] in dur:
                acc[key]["_us"].append(dur[r["Dispatch_Id"]])
    out = {"_source": "rocprofv3 --pmc <counters> --kernel-trace (separate passes, scripts/pmc_sq.sh) of "
                      "scripts/micro/convbench --iters 3 --rounds 1 --ops fdw --filter <layer> (C ABI, batch 64, bf16); "
                      "values are averages per launch; launch durations are those of the profiled passes"}
    for (layer, k, grid), c in sorted(acc.items()):
        v = {n: sum(x) / len(x) for n, x in c.items()}
        e = {"layer": layer, "kernel": k, "grid": int(grid), "launches_sampled": len(c.get("_us", [])) or max(len(x) for x in c.values()),
             "avg_us_profiled": round(v.get("_us", 0.0), 1), "counters": {n: round(x, 1) for n, x in v.items() if n != "_us"}}
        d = {}
        g = v.get
        if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("SQ_BUSY_CU_CYCLES"):
            d["mfma_busy_frac"] = round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (4 * g("SQ_BUSY_CU_CYCLES")), 4)
        if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("SQ_BUSY_CYCLES"):
            d["mfma_busy_vs_sq_busy"] = round(g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_BUSY_CYCLES"), 4)
        if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("SQ_INSTS_MFMA"):
            d["cycles_per_mfma"] = round(g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_INSTS_MFMA"), 2)
        if g("SQ_WAVE_CYCLES"):
            for n, src in (("wait_frac", "SQ_WAIT_ANY"), ("issue_stall_frac", "SQ_WAIT_INST_ANY"), ("active_frac", "SQ_ACTIVE_INST_ANY")):
                if g(src) is not None:
                    d[n] = round(g(src) / g("SQ_WAVE_CYCLES"), 4)
        if g("SQ_LDS_IDX_ACTIVE"):
            d["lds_conflict_frac"] = round(g("SQ_LDS_BANK_CONFLICT", 0.0) / g("SQ_LDS_IDX_ACTIVE"), 4)
        if g("GRBM_GUI_ACTIVE") and v.get("_us"):
            d["clock_ghz"] = round(g("GRBM_GUI_ACTIVE") / 8 / (v["_us"] * 1e3), 3)
            if g("SQ_VALU_MFMA_BUSY_CYCLES"):
                # matrix-pipe busy cycles (summed over the chip's 1024 SIMDs) / SIMD-cycles the launch lasted
                d["mfma_busy_of_launch"] = round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * v["_us"] * 1e3 * d["clock_ghz"]), 4)
        e["derived"] = d
        out.setdefault("kernels", []).append(e)
    # merge entries of the same (layer, kernel, grid) coming from different passes: done by the dict above (one key)
    json.dump(out, open(out_path, "w"), indent=1)
    for e in out.get("kernels", []):
        print(f"{e['layer']:6s} {e['kernel'][:46]:46s} grid {e['grid']:>7d} {e['avg_us_profiled']:8.1f} us  {e['derived']}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
